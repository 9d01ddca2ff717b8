// Levelised functional sweep on bf16x3 split-precision MFMA (mgv_x3.h): same operator, arguments and
// phases as func_level.hip, whose header describes the math.  Differences: the zbar tile and the gate
// gradients live in LDS as bf16 hi/lo planes, the three dense products (recompute, dgrad, wgrad) run on
// v_mfma_f32_16x16x32_bf16, the wgrad reads its operands transposed (ds_read_b64_tr_b16) from the same
// row-major planes, and the weights arrive pre-split in MFMA fragment order:
//   per slot:  Wvc_hi | Wvc_lo   ([3H][2H], blocks (row tile, k-step))   forward / recompute B operands
//              WvcT_hi | WvcT_lo ([2H][3H])                               dgrad B operands
#include "mgv_x3.h"
#include "../../include/mgvae_hip.h"

namespace mgv {

constexpr uint8_t kNoGateX = 255;

struct LevelX3Args {
    int64_t N;
    int T;
    const int32_t* order; const int32_t* tile_start; const int32_t* tile_count; const int32_t* tile_slot;
    int tile_begin;
    const int32_t* in_ptr; const int32_t* in_src;
    const float* hs; float* hf;
    const float* attn_u;   // [T][2H]
    const __bf16* wpack;   // [T][4][6H^2]
    const float* bvc; const float* bih; const float* bhh;   // [T][3H]
    const int32_t* out_ptr; const int32_t* out_dst; const int32_t* out_slot; const uint8_t* gslot;
    const float* ghf; float* ghs; float* dzb; float* alpha; float* dsc;
    float* d_attn_u; float* dWvc; float* dbvc; float* dbih; float* dbhh;
};

template <int H>
struct LvlSmem {
    using S = WaveSplit<H>;
    static constexpr int LDZP = 2 * H + 8;                    // bf16 elements per zbar plane row
    static constexpr int LDGP = H + 8;                        // bf16 elements per gate-gradient plane row
    static constexpr int LDZF = 2 * H + 4;                    // fp32 d(zbar) tile row
    static constexpr int ZPB = kTileRows * LDZP * 2;          // bytes of one zbar plane
    static constexpr int GPB = kTileRows * LDGP * 2;
    static constexpr int o_zhi = 0;
    static constexpr int o_zlo = o_zhi + ZPB;
    // forward: output tile fp32 [64][H+4] reuses the zbar planes
    static constexpr int o_small_f = o_zlo + ZPB;
    static constexpr int SMALL_F = 2 * H + 9 * H + 3 * kTileRows + kTileRows;     // u, bvc/bih/bhh, sa/m/inv, node
    static constexpr int fwd_bytes = o_small_f + SMALL_F * 4;
    // backward: region R after the planes: dh fp32, then gate-gradient planes, then d(zbar) fp32
    static constexpr int o_r = o_zlo + ZPB;
    static constexpr int R_BYTES = kTileRows * LDZF * 4;
    static_assert(2 * GPB <= R_BYTES && kTileRows * (H + 4) * 4 <= R_BYTES, "region R");
    static constexpr int o_small_b = o_r + R_BYTES;
    static constexpr int o_acc = o_small_b + SMALL_F * 4;     // gu[2H], dbvc, dbih, dbhh [3H each]
    static constexpr int bwd_bytes = o_acc + (2 * H + 9 * H) * 4;
};

struct LvlSmall { float* u; float* bvc; float* bih; float* bhh; float* sa; float* m; float* inv; int* node; };

template <int H>
__device__ __forceinline__ LvlSmall lvl_small(const LevelX3Args& a, int g, float* base) {
    LvlSmall v;
    v.u = base; v.bvc = v.u + 2 * H; v.bih = v.bvc + 3 * H; v.bhh = v.bih + 3 * H;
    v.sa = v.bhh + 3 * H; v.m = v.sa + kTileRows; v.inv = v.m + kTileRows;
    v.node = reinterpret_cast<int*>(v.inv + kTileRows);
    for (int i = threadIdx.x; i < 2 * H; i += kThreads) v.u[i] = a.attn_u[(int64_t)g * 2 * H + i];
    for (int i = threadIdx.x; i < 3 * H; i += kThreads) {
        v.bvc[i] = a.bvc[(int64_t)g * 3 * H + i]; v.bih[i] = a.bih[(int64_t)g * 3 * H + i]; v.bhh[i] = a.bhh[(int64_t)g * 3 * H + i];
    }
    return v;
}

// attention over the in-edges of `node` (online softmax), as in func_level.hip
template <int H>
__device__ __forceinline__ void attn_row_x(const LevelX3Args& a, int64_t node, const float4& us, const float4& uf, int lr,
                                           float& m, float& inv, float4& zs, float4& zf) {
    constexpr int LPR = H / 4;
    const int e0 = a.in_ptr[node], e1 = a.in_ptr[node + 1];
    m = -INFINITY;
    float S = 0.f;
    zs = zero4(); zf = zero4();
    for (int e = e0; e < e1; ++e) {
        const int64_t j = a.in_src[e];
        const float4 xs = ld4(a.hs + j * H + 4 * lr), xf = ld4(a.hf + j * H + 4 * lr);
        const float sc = group_sum<LPR>(dot4(us, xs) + dot4(uf, xf));
        const float mn = fmaxf(m, sc);
        const float corr = __expf(m - mn), w = __expf(sc - mn);
        S = S * corr + w;
        zs = fma4(w, xs, scale4(corr, zs));
        zf = fma4(w, xf, scale4(corr, zf));
        m = mn;
    }
    inv = 1.0f / (S + 1e-16f);
    zs = scale4(inv, zs); zf = scale4(inv, zf);
    if (e1 == e0) m = 0.f;
}

template <int H>
__device__ __forceinline__ void pull_row_x(const LevelX3Args& a, int64_t node, int lr, float4& gs, float4& gf) {
    const int e0 = a.out_ptr[node], e1 = a.out_ptr[node + 1];
    gs = zero4(); gf = zero4();
    for (int e = e0; e < e1; ++e) {
        const int64_t c = a.out_dst[e];
        const int gc = a.gslot[c];
        if (gc == kNoGateX) continue;
        const int sl = a.out_slot[e];
        const float al = a.alpha[sl], ds = a.dsc[sl];
        const float* dz = a.dzb + c * 2 * H;
        const float* u = a.attn_u + (int64_t)gc * 2 * H;
        gs = fma4(al, ld4(dz + 4 * lr), fma4(ds, ld4(u + 4 * lr), gs));
        gf = fma4(al, ld4(dz + H + 4 * lr), fma4(ds, ld4(u + H + 4 * lr), gf));
    }
}

// gate pre-activations (r, z, n blocks) = zbar[64 x 2H] * Wvc_g^T from the split planes
template <int H>
__device__ __forceinline__ void lvl_gemm_x3(const __bf16* wslot, const __bf16* z_hi, const __bf16* z_lo,
                                            f32x4 (&ar)[WaveSplit<H>::RTW], f32x4 (&az)[WaveSplit<H>::RTW], f32x4 (&an)[WaveSplit<H>::RTW]) {
    using S = WaveSplit<H>;
    static_assert(S::HCW == 1, "one hidden-column tile per wave");
    constexpr int LDZP = 2 * H + 8, BLK = 6 * H * H, KS = 2 * H / 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
#pragma unroll
    for (int i = 0; i < S::RTW; ++i) { ar[i] = f32x4{0.f, 0.f, 0.f, 0.f}; az[i] = ar[i]; an[i] = ar[i]; }
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
        bf16x8 xh[S::RTW], xl[S::RTW];
#pragma unroll
        for (int i = 0; i < S::RTW; ++i) {
            const int off = ((wr * S::RTW + i) * 16 + r) * LDZP + 32 * ks + 8 * q;
            xh[i] = ldfrag(z_hi + off); xl[i] = ldfrag(z_lo + off);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int wo = ((g * (H / 16) + wc) * KS + ks) * 512 + lane * 8;
            const bf16x8 bh = ldfrag(wslot + wo), bl = ldfrag(wslot + BLK + wo);
#pragma unroll
            for (int i = 0; i < S::RTW; ++i) {
                if (g == 0) mma_x3(ar[i], xh[i], xl[i], bh, bl);
                if (g == 1) mma_x3(az[i], xh[i], xl[i], bh, bl);
                if (g == 2) mma_x3(an[i], xh[i], xl[i], bh, bl);
            }
        }
    }
}

template <int H>
__global__ __launch_bounds__(kThreads) void k_level_fwd_x3(LevelX3Args a) {
    using S = WaveSplit<H>;
    using M = LvlSmem<H>;
    constexpr int LDZP = M::LDZP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* z_hi = reinterpret_cast<__bf16*>(smem_raw + M::o_zhi);
    __bf16* z_lo = reinterpret_cast<__bf16*>(smem_raw + M::o_zlo);
    float* s_o = reinterpret_cast<float*>(smem_raw + M::o_zhi);       // output tile, after the MFMAs
    const int tile = a.tile_begin + blockIdx.x;
    const int start = a.tile_start[tile], count = a.tile_count[tile], g = a.tile_slot[tile];
    const LvlSmall sv = lvl_small<H>(a, g, reinterpret_cast<float*>(smem_raw + M::o_small_f));
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
    const int grp = tid / S::LPR, lr = tid % S::LPR;
    __syncthreads();
    const float4 us = ld4(sv.u + 4 * lr), uf = ld4(sv.u + H + 4 * lr);
    for (int row = grp; row < kTileRows; row += S::GROUPS) {
        float4 zs = zero4(), zf = zero4();
        float sa = 0.f;
        int node = -1;
        if (row < count) {
            node = a.order[start + row];
            float m, inv;
            attn_row_x<H>(a, node, us, uf, lr, m, inv, zs, zf);
            sa = a.in_ptr[node + 1] > a.in_ptr[node] ? 1.0f : 0.0f;
        }
        bf16x4 hi, lo;
        split4(zs, hi, lo);
        st_bf4(z_hi + row * LDZP + 4 * lr, hi); st_bf4(z_lo + row * LDZP + 4 * lr, lo);
        split4(zf, hi, lo);
        st_bf4(z_hi + row * LDZP + H + 4 * lr, hi); st_bf4(z_lo + row * LDZP + H + 4 * lr, lo);
        if (lr == 0) { sv.sa[row] = sa; sv.node[row] = node; }
    }
    __syncthreads();
    f32x4 ar[S::RTW], az[S::RTW], an[S::RTW];
    lvl_gemm_x3<H>(a.wpack + (int64_t)g * 4 * 6 * H * H, z_hi, z_lo, ar, az, an);
    __syncthreads();                   // s_o overlays the planes
    {
        const int col = wc * 16 + r;
        const float bvr = sv.bvc[col], bvz = sv.bvc[H + col], bvn = sv.bvc[2 * H + col];
        const float cr = sv.bih[col] + sv.bhh[col], cz = sv.bih[H + col] + sv.bhh[H + col], cn = sv.bih[2 * H + col];
        const float bhn = sv.bhh[2 * H + col];
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                const float sa = sv.sa[row];
                const float rr = sigmoidf_(ar[i][e] + sa * bvr + cr);
                const float zz = sigmoidf_(az[i][e] + sa * bvz + cz);
                const float nn = tanhf_(an[i][e] + sa * bvn + cn + rr * bhn);
                s_o[row * S::LD + col] = (1.0f - zz) * nn;        // h0 = 0
            }
    }
    __syncthreads();
    for (int row = grp; row < count; row += S::GROUPS)
        st4(a.hf + (int64_t)sv.node[row] * H + 4 * lr, ld4(s_o + row * S::LD + 4 * lr));
}

__device__ __forceinline__ void colsum_lds_lx(float v, float* dst) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if ((threadIdx.x & 63) < 16) atomicAdd(dst, v);
}

template <int H>
__global__ __launch_bounds__(kThreads) void k_level_bwd_x3(LevelX3Args a) {
    using S = WaveSplit<H>;
    using S2 = WaveSplit<2 * H>;
    using M = LvlSmem<H>;
    constexpr int LDZP = M::LDZP, LDGP = M::LDGP, LDZF = M::LDZF, BLK = 6 * H * H;
    constexpr int TI = H / 16, TJ = 2 * H / 16, TT = TI * TJ, TPW = TT / 4;
    static_assert(TT % 4 == 0, "wgrad tiles must split over 4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* z_hi = reinterpret_cast<__bf16*>(smem_raw + M::o_zhi);
    __bf16* z_lo = reinterpret_cast<__bf16*>(smem_raw + M::o_zlo);
    float* s_dh = reinterpret_cast<float*>(smem_raw + M::o_r);
    __bf16* d_hi = reinterpret_cast<__bf16*>(smem_raw + M::o_r);
    __bf16* d_lo = d_hi + kTileRows * LDGP;
    float* s_dz = reinterpret_cast<float*>(smem_raw + M::o_r);
    const int tile = a.tile_begin + blockIdx.x;
    const int start = a.tile_start[tile], count = a.tile_count[tile], g = a.tile_slot[tile];
    const LvlSmall sv = lvl_small<H>(a, g, reinterpret_cast<float*>(smem_raw + M::o_small_b));
    float* s_gu = reinterpret_cast<float*>(smem_raw + M::o_acc);
    float* s_dbvc = s_gu + 2 * H; float* s_dbih = s_dbvc + 3 * H; float* s_dbhh = s_dbih + 3 * H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
    const int grp = tid / S::LPR, lr = tid % S::LPR;
    for (int i = tid; i < 2 * H + 9 * H; i += kThreads) s_gu[i] = 0.f;
    __syncthreads();
    const float4 us = ld4(sv.u + 4 * lr), uf = ld4(sv.u + H + 4 * lr);
    // ---- 0/1. pull dL/dhf (and finish dL/dhs) of the tile's nodes, recompute their attention
    for (int row = grp; row < kTileRows; row += S::GROUPS) {
        float4 zs = zero4(), zf = zero4(), dh = zero4();
        float sa = 0.f, m = 0.f, inv = 0.f;
        int node = -1;
        if (row < count) {
            node = a.order[start + row];
            float4 gs, gf;
            pull_row_x<H>(a, node, lr, gs, gf);
            dh = add4(gf, ld4(a.ghf + (int64_t)node * H + 4 * lr));
            float* gp = a.ghs + (int64_t)node * H + 4 * lr;
            st4(gp, add4(ld4(gp), gs));
            attn_row_x<H>(a, node, us, uf, lr, m, inv, zs, zf);
            sa = a.in_ptr[node + 1] > a.in_ptr[node] ? 1.0f : 0.0f;
        }
        bf16x4 hi, lo;
        split4(zs, hi, lo);
        st_bf4(z_hi + row * LDZP + 4 * lr, hi); st_bf4(z_lo + row * LDZP + 4 * lr, lo);
        split4(zf, hi, lo);
        st_bf4(z_hi + row * LDZP + H + 4 * lr, hi); st_bf4(z_lo + row * LDZP + H + 4 * lr, lo);
        st4(s_dh + row * S::LD + 4 * lr, dh);
        if (lr == 0) { sv.sa[row] = sa; sv.m[row] = m; sv.inv[row] = inv; sv.node[row] = node; }
    }
    __syncthreads();
    // ---- 2. recompute gates
    const __bf16* wslot = a.wpack + (int64_t)g * 4 * BLK;
    f32x4 ar[S::RTW], az[S::RTW], an[S::RTW];
    lvl_gemm_x3<H>(wslot, z_hi, z_lo, ar, az, an);
    // ---- 3. GRU backward (h0 = 0: hf = (1-z) n, gh = b_hh); ar/az/an become da_r/da_z/da_n
    {
        const int col = wc * 16 + r;
        const float bvr = sv.bvc[col], bvz = sv.bvc[H + col], bvn = sv.bvc[2 * H + col];
        const float cr = sv.bih[col] + sv.bhh[col], cz = sv.bih[H + col] + sv.bhh[H + col], cn = sv.bih[2 * H + col];
        const float bhn = sv.bhh[2 * H + col];
        float b_r = 0.f, b_z = 0.f, b_n = 0.f, v_r = 0.f, v_z = 0.f, v_n = 0.f, h_n = 0.f;
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                const float sa = sv.sa[row];
                const float rr = sigmoidf_(ar[i][e] + sa * bvr + cr);
                const float zz = sigmoidf_(az[i][e] + sa * bvz + cz);
                const float nn = tanhf_(an[i][e] + sa * bvn + cn + rr * bhn);
                const float dh = s_dh[row * S::LD + col];
                const float dan = dh * (1.0f - zz) * (1.0f - nn * nn);
                const float daz = -dh * nn * zz * (1.0f - zz);
                const float dar = dan * bhn * rr * (1.0f - rr);
                ar[i][e] = dar; az[i][e] = daz; an[i][e] = dan;
                b_r += dar; b_z += daz; b_n += dan; h_n += dan * rr;
                v_r += sa * dar; v_z += sa * daz; v_n += sa * dan;
            }
        colsum_lds_lx(b_r, s_dbih + col); colsum_lds_lx(b_z, s_dbih + H + col); colsum_lds_lx(b_n, s_dbih + 2 * H + col);
        colsum_lds_lx(b_r, s_dbhh + col); colsum_lds_lx(b_z, s_dbhh + H + col); colsum_lds_lx(h_n, s_dbhh + 2 * H + col);
        colsum_lds_lx(v_r, s_dbvc + col); colsum_lds_lx(v_z, s_dbvc + H + col); colsum_lds_lx(v_n, s_dbvc + 2 * H + col);
    }
    // ---- 4. three passes: d(zbar) += dG_p * Wvc[p]  and  dWvc[p] += dG_p^T * zbar
    const int wc2 = w % S2::WPC, wr2 = w / S2::WPC;
    f32x4 dz[S2::RTW][S2::HCW];
#pragma unroll
    for (int i = 0; i < S2::RTW; ++i)
#pragma unroll
        for (int j = 0; j < S2::HCW; ++j) dz[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* dWvc_g = a.dWvc + (int64_t)g * 3 * H * 2 * H;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        __syncthreads();               // readers of region R: phase 3 (dh), or the previous pass
        {
            const int col = wc * 16 + r;
#pragma unroll
            for (int i = 0; i < S::RTW; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                    const float v = p == 0 ? ar[i][e] : p == 1 ? az[i][e] : an[i][e];
                    __bf16 hh, ll;
                    split_bf16(v, hh, ll);
                    d_hi[row * LDGP + col] = hh; d_lo[row * LDGP + col] = ll;
                }
        }
        __syncthreads();
#pragma unroll 1
        for (int ks = 0; ks < H / 32; ++ks) {
            bf16x8 xh[S2::RTW], xl[S2::RTW];
#pragma unroll
            for (int i = 0; i < S2::RTW; ++i) {
                const int off = ((wr2 * S2::RTW + i) * 16 + r) * LDGP + 32 * ks + 8 * q;
                xh[i] = ldfrag(d_hi + off); xl[i] = ldfrag(d_lo + off);
            }
#pragma unroll
            for (int j = 0; j < S2::HCW; ++j) {
                const int ct2 = wc2 * S2::HCW + j;
                const int wo = ((ct2 * 3 + p) * (H / 32) + ks) * 512 + lane * 8;
                const bf16x8 bh = ldfrag(wslot + 2 * BLK + wo), bl = ldfrag(wslot + 3 * BLK + wo);
#pragma unroll
                for (int i = 0; i < S2::RTW; ++i) mma_x3(dz[i][j], xh[i], xl[i], bh, bl);
            }
        }
        // weight gradient of this gate block: both operands read transposed from the row-major planes
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tl = w * TPW + t;
            const int it = tl / TJ, jt = tl % TJ;
            f32x4 gw = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < kTileRows / 32; ++ks)
                mma_x3(gw, ldfrag_tr(d_hi, LDGP, 32 * ks, it * 16), ldfrag_tr(d_lo, LDGP, 32 * ks, it * 16),
                       ldfrag_tr(z_hi, LDZP, 32 * ks, jt * 16), ldfrag_tr(z_lo, LDZP, 32 * ks, jt * 16));
#pragma unroll
            for (int e = 0; e < 4; ++e)
                atomicAdd(dWvc_g + (int64_t)(p * H + it * 16 + q * 4 + e) * 2 * H + jt * 16 + r, gw[e]);
        }
    }
    // ---- 5. d(zbar) tile to LDS (fp32, row layout for the attention backward); it overlays the dG planes
    __syncthreads();
#pragma unroll
    for (int i = 0; i < S2::RTW; ++i)
#pragma unroll
        for (int j = 0; j < S2::HCW; ++j) {
            const int col = (wc2 * S2::HCW + j) * 16 + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) s_dz[((wr2 * S2::RTW + i) * 16 + q * 4 + e) * LDZF + col] = dz[i][j][e];
        }
    __syncthreads();
    // ---- 6. attention backward per in-edge
    float4 gus = zero4(), guf = zero4();
    for (int row = grp; row < count; row += S::GROUPS) {
        const int64_t node = sv.node[row];
        const float4 dzs = ld4(s_dz + row * LDZF + 4 * lr), dzf = ld4(s_dz + row * LDZF + H + 4 * lr);
        const bf16x4 zsh = *reinterpret_cast<const bf16x4*>(z_hi + row * LDZP + 4 * lr), zsl = *reinterpret_cast<const bf16x4*>(z_lo + row * LDZP + 4 * lr);
        const bf16x4 zfh = *reinterpret_cast<const bf16x4*>(z_hi + row * LDZP + H + 4 * lr), zfl = *reinterpret_cast<const bf16x4*>(z_lo + row * LDZP + H + 4 * lr);
        const float4 zs = make_float4((float)zsh[0] + (float)zsl[0], (float)zsh[1] + (float)zsl[1], (float)zsh[2] + (float)zsl[2], (float)zsh[3] + (float)zsl[3]);
        const float4 zf = make_float4((float)zfh[0] + (float)zfl[0], (float)zfh[1] + (float)zfl[1], (float)zfh[2] + (float)zfl[2], (float)zfh[3] + (float)zfl[3]);
        st4(a.dzb + node * 2 * H + 4 * lr, dzs);
        st4(a.dzb + node * 2 * H + H + 4 * lr, dzf);
        const float ci = group_sum<S::LPR>(dot4(dzs, zs) + dot4(dzf, zf));
        const float m = sv.m[row], inv = sv.inv[row];
        const int e0 = a.in_ptr[node], e1 = a.in_ptr[node + 1];
        for (int e = e0; e < e1; ++e) {
            const int64_t j = a.in_src[e];
            const float4 xs = ld4(a.hs + j * H + 4 * lr), xf = ld4(a.hf + j * H + 4 * lr);
            const float sc = group_sum<S::LPR>(dot4(us, xs) + dot4(uf, xf));
            const float t = group_sum<S::LPR>(dot4(dzs, xs) + dot4(dzf, xf));
            const float al = __expf(sc - m) * inv;
            const float ds = al * (t - ci);
            if (lr == 0) { a.alpha[e] = al; a.dsc[e] = ds; }
            gus = fma4(ds, xs, gus);
            guf = fma4(ds, xf, guf);
        }
    }
    atomicAdd(&s_gu[4 * lr + 0], gus.x); atomicAdd(&s_gu[4 * lr + 1], gus.y); atomicAdd(&s_gu[4 * lr + 2], gus.z); atomicAdd(&s_gu[4 * lr + 3], gus.w);
    atomicAdd(&s_gu[H + 4 * lr + 0], guf.x); atomicAdd(&s_gu[H + 4 * lr + 1], guf.y); atomicAdd(&s_gu[H + 4 * lr + 2], guf.z); atomicAdd(&s_gu[H + 4 * lr + 3], guf.w);
    __syncthreads();
    for (int i = tid; i < 2 * H; i += kThreads) atomicAdd(a.d_attn_u + (int64_t)g * 2 * H + i, s_gu[i]);
    for (int i = tid; i < 3 * H; i += kThreads) {
        atomicAdd(a.dbvc + (int64_t)g * 3 * H + i, s_dbvc[i]);
        atomicAdd(a.dbih + (int64_t)g * 3 * H + i, s_dbih[i]);
        atomicAdd(a.dbhh + (int64_t)g * 3 * H + i, s_dbhh[i]);
    }
}

template <int H>
__global__ __launch_bounds__(kThreads) void k_level_pull_inactive_x3(LevelX3Args a) {
    constexpr int LPR = H / 4;
    const int lr = threadIdx.x % LPR;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / LPR);
    for (int64_t node = (int64_t)blockIdx.x * (kThreads / LPR) + threadIdx.x / LPR; node < a.N; node += stride) {
        if (a.gslot[node] != kNoGateX) continue;
        float4 gs, gf;
        pull_row_x<H>(a, node, lr, gs, gf);
        float* gp = a.ghs + node * H + 4 * lr;
        st4(gp, add4(ld4(gp), gs));
    }
}

template <int H>
int launch_level_x3(bool bwd, const LevelX3Args& a, int ntiles, hipStream_t st) {
    using M = LvlSmem<H>;
    if (bwd) {
        static bool set_b = false;
        if (!set_b) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_level_bwd_x3<H>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set_b = true; }
        hipLaunchKernelGGL(k_level_bwd_x3<H>, dim3(ntiles), dim3(kThreads), M::bwd_bytes, st, a);
    } else {
        static bool set_f = false;
        if (!set_f) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_level_fwd_x3<H>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set_f = true; }
        hipLaunchKernelGGL(k_level_fwd_x3<H>, dim3(ntiles), dim3(kThreads), M::fwd_bytes, st, a);
    }
    MGV_LAUNCH_RET();
}

}  // namespace mgv

extern "C" int mgv_func_sweep_fwd_x3(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                     const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                                     const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src, const float* hs,
                                     float* hf, const float* attn_u, const void* wpack_bf16, const float* bvc, const float* bih,
                                     const float* bhh, void* stream) {
    MGV_CHECK_ARG(N >= 0 && T >= 1 && num_levels >= 0 && level_tile_ptr_host && hs && hf && attn_u && wpack_bf16 && bvc && bih && bhh && in_ptr);
    mgv::LevelX3Args a{};
    a.N = N; a.T = T; a.order = order; a.tile_start = tile_start; a.tile_count = tile_count; a.tile_slot = tile_slot;
    a.in_ptr = in_ptr; a.in_src = in_src; a.hs = hs; a.hf = hf; a.attn_u = attn_u; a.wpack = static_cast<const __bf16*>(wpack_bf16);
    a.bvc = bvc; a.bih = bih; a.bhh = bhh;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int lv = 1; lv < num_levels; ++lv) {
        const int t0 = level_tile_ptr_host[lv], t1 = level_tile_ptr_host[lv + 1];
        if (t1 <= t0) continue;
        MGV_CHECK_ARG(order && tile_start && tile_count && tile_slot && in_src);
        a.tile_begin = t0;
        int rc;
        switch (H) {
            case 32: rc = mgv::launch_level_x3<32>(false, a, t1 - t0, st); break;
            case 64: rc = mgv::launch_level_x3<64>(false, a, t1 - t0, st); break;
            default: return MGV_EUNSUPPORTED;
        }
        if (rc != MGV_OK) return rc;
    }
    return MGV_OK;
}

extern "C" int mgv_func_sweep_bwd_x3(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                     const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                                     const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src,
                                     const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_slot,
                                     const uint8_t* gslot, const float* hs, const float* hf, const float* attn_u,
                                     const void* wpack_bf16, const float* bvc, const float* bih, const float* bhh,
                                     const float* ghf, float* ghs, float* dzb, float* alpha, float* dsc, float* d_attn_u,
                                     float* dWvc, float* dbvc, float* dbih, float* dbhh, void* stream) {
    MGV_CHECK_ARG(N >= 0 && T >= 1 && num_levels >= 0 && level_tile_ptr_host && hs && hf && attn_u && wpack_bf16 && bvc && bih && bhh);
    MGV_CHECK_ARG(in_ptr && out_ptr && gslot && ghf && ghs && dzb && d_attn_u && dWvc && dbvc && dbih && dbhh);
    if (N == 0) return MGV_OK;
    mgv::LevelX3Args a{};
    a.N = N; a.T = T; a.order = order; a.tile_start = tile_start; a.tile_count = tile_count; a.tile_slot = tile_slot;
    a.in_ptr = in_ptr; a.in_src = in_src; a.hs = hs; a.hf = const_cast<float*>(hf); a.attn_u = attn_u;
    a.wpack = static_cast<const __bf16*>(wpack_bf16); a.bvc = bvc; a.bih = bih; a.bhh = bhh;
    a.out_ptr = out_ptr; a.out_dst = out_dst; a.out_slot = out_slot; a.gslot = gslot;
    a.ghf = ghf; a.ghs = ghs; a.dzb = dzb; a.alpha = alpha; a.dsc = dsc; a.d_attn_u = d_attn_u; a.dWvc = dWvc; a.dbvc = dbvc;
    a.dbih = dbih; a.dbhh = dbhh;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int lv = num_levels - 1; lv >= 1; --lv) {
        const int t0 = level_tile_ptr_host[lv], t1 = level_tile_ptr_host[lv + 1];
        if (t1 <= t0) continue;
        MGV_CHECK_ARG(order && tile_start && tile_count && tile_slot && in_src && out_dst && out_slot && alpha && dsc);
        a.tile_begin = t0;
        int rc;
        switch (H) {
            case 32: rc = mgv::launch_level_x3<32>(true, a, t1 - t0, st); break;
            case 64: rc = mgv::launch_level_x3<64>(true, a, t1 - t0, st); break;
            default: return MGV_EUNSUPPORTED;
        }
        if (rc != MGV_OK) return rc;
    }
    const int rows_per_block = mgv::kThreads / (H / 4);
    const int grid = mgv::grid_for((N + rows_per_block - 1) / rows_per_block, 8);
    switch (H) {
        case 32: hipLaunchKernelGGL(mgv::k_level_pull_inactive_x3<32>, dim3(grid), dim3(mgv::kThreads), 0, st, a); break;
        case 64: hipLaunchKernelGGL(mgv::k_level_pull_inactive_x3<64>, dim3(grid), dim3(mgv::kThreads), 0, st, a); break;
        default: return MGV_EUNSUPPORTED;
    }
    MGV_LAUNCH_RET();
}
