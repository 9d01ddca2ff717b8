// Levelised functional sweep: per-gate-type attention aggregation (TFMlpAggr) + GRU update of hf,
// one launch per logic level, one workgroup per <=64-node tile of a single gate type.
//
// Reference (dg_ae_model_aig.py:70-97 and the mig/xag/xmg siblings, arch/tfmlp.py:38-46), per node i
// of gate g with in-edges j->i and x_j = [hs_j, hf_j]:
//     a_j = attn([Wq x_i + bq, Wk x_j + bk]);  alpha = softmax_j(a);  msg_i = sum_j alpha_j (Wv x_j + bv)
//     hf_i = GRU_g(msg_i, hf_i)                      (num_rounds = 1  =>  hf_i = 0 before its update)
// Restated per NODE: the q term and every bias inside `a` are constant over a softmax segment and
// cancel, so a_j ~ u_g . x_j with u_g = Wk^T w_attn[H:];  msg_i = Wv (sum_j alpha_j x_j) + bv sum_j alpha_j;
// the message Linear is folded into the GRU input projection (Wvc = W_ih Wv, bvc = W_ih bv).  Host code
// composes u_g / Wvc / bvc (tiny matrices; autograd differentiates the composition).
//
// Backward walks the levels in reverse.  No scatter: a node PULLS the gradient of its hf/hs rows from
// its consumers (out-CSR), which left alpha, d(score) per in-edge and d(zbar) per node behind.
#include "mgv_common.h"
#include "../../include/mgvae_hip.h"

namespace mgv {

constexpr uint8_t kNoGate = 255;

struct LevelArgs {
    int64_t N;
    int T;
    const int32_t* order; const int32_t* tile_start; const int32_t* tile_count; const int32_t* tile_slot;
    int tile_begin;
    const int32_t* in_ptr; const int32_t* in_src;
    const float* hs; float* hf;
    const float* attn_u;   // [T][2H]
    const float* Wvc;      // [T][3H][2H]
    const float* bvc; const float* bih; const float* bhh;   // [T][3H]
    // backward
    const float* WvcT;     // [T][2H][3H]
    const int32_t* out_ptr; const int32_t* out_dst; const int32_t* out_slot; const uint8_t* gslot;
    const float* ghf;      // [N][H] gradient wrt hf from the losses
    float* ghs;            // [N][H] += gradient wrt hs from the sweep
    float* dzb;            // [N][2H] d(zbar) per updated node
    float* alpha; float* dsc;   // [E] per in-edge (in-CSR order)
    float* d_attn_u; float* dWvc; float* dbvc; float* dbih; float* dbhh;
    // rounds >= 2 (dg_ae_model_aig.py:70-97 with num_rounds > 1; same contract as LevelX3Args): gh[node][3H] = W_hh h_prev + b_hh of
    // the node's own aggregator (bhh = 0 here), hprev[node][H]; backward leaves d(gh) and d(h_prev) = dh * z.  NULL in round 1.
    const float* gh; const float* hprev; float* dgh; float* ghprev;
};

template <int H, bool BWD = true>
struct LevelSmem {
    using S = WaveSplit<H>;
    static constexpr int LDZ = 2 * H + 4;
    static constexpr int off_z = 0;                                  // zbar tile [64][2H+4]
    // second tile region [64][2H+4]: backward keeps dh, then the dG tiles ([64][H+4]) in it and finally
    // overlays d(zbar) ([64][2H+4]) on it; forward needs no second region (its output tile reuses off_z)
    static constexpr int off_o = off_z + kTileRows * LDZ;
    static constexpr int off_u = BWD ? off_o + kTileRows * LDZ : off_o;    // forward: no second region
    static constexpr int off_bvc = off_u + 2 * H;
    static constexpr int off_bih = off_bvc + 3 * H;
    static constexpr int off_bhh = off_bih + 3 * H;
    static constexpr int off_sa = off_bhh + 3 * H;                   // sum of alphas per row
    static constexpr int off_m = off_sa + kTileRows;                 // softmax max per row
    static constexpr int off_inv = off_m + kTileRows;                // 1/(S+1e-16) per row
    static constexpr int off_node = off_inv + kTileRows;             // node id per row (int)
    static constexpr int fwd_floats = off_node + kTileRows;
    static constexpr int off_dz = off_o;                             // d(zbar) tile [64][2H+4] over the dG region
    static constexpr int off_gu = fwd_floats;                        // [2H]
    static constexpr int off_dbvc = off_gu + 2 * H;
    static constexpr int off_dbih = off_dbvc + 3 * H;
    static constexpr int off_dbhh = off_dbih + 3 * H;
    static constexpr int bwd_floats = off_dbhh + 3 * H;
};

// attention over the in-edges of `node`, online softmax; lanes of a row group hold float4 slices
template <int H>
__device__ __forceinline__ void attn_row(const LevelArgs& a, int64_t node, const float4& us, const float4& uf, int lr,
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
    inv = 1.0f / (S + 1e-16f);       // torch_geometric softmax: e / (sum e + 1e-16)
    zs = scale4(inv, zs); zf = scale4(inv, zf);
    // sum of alphas = S * inv (1 for any node with an in-edge, 0 otherwise); returned through inv and S:
    // the caller recomputes S*inv as (deg>0)
    if (e1 == e0) { m = 0.f; }
}

template <int H, bool BWD>
__device__ __forceinline__ void level_stage_vectors(const LevelArgs& a, int g, float* smem) {
    using M = LevelSmem<H, BWD>;
    for (int i = threadIdx.x; i < 2 * H; i += kThreads) smem[M::off_u + i] = a.attn_u[(int64_t)g * 2 * H + i];
    for (int i = threadIdx.x; i < 3 * H; i += kThreads) {
        smem[M::off_bvc + i] = a.bvc[(int64_t)g * 3 * H + i];
        smem[M::off_bih + i] = a.bih[(int64_t)g * 3 * H + i];
        smem[M::off_bhh + i] = a.bhh[(int64_t)g * 3 * H + i];
    }
}

// gi (three gate blocks) = zbar[64 x 2H] * Wvc_g^T, accumulator layout
template <int H>
__device__ __forceinline__ void level_gemm(const float* Wvc_g, const float* s_z,
                                           f32x4 (&ar)[WaveSplit<H>::RTW][WaveSplit<H>::HCW],
                                           f32x4 (&az)[WaveSplit<H>::RTW][WaveSplit<H>::HCW],
                                           f32x4 (&an)[WaveSplit<H>::RTW][WaveSplit<H>::HCW]) {
    using S = WaveSplit<H>;
    constexpr int LDZ = LevelSmem<H>::LDZ;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
#pragma unroll
    for (int i = 0; i < S::RTW; ++i)
#pragma unroll
        for (int j = 0; j < S::HCW; ++j) { ar[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; az[i][j] = ar[i][j]; an[i][j] = ar[i][j]; }
#pragma unroll 1
    for (int kb = 0; kb < 2 * H; kb += 16) {
        float4 xz[S::RTW];
#pragma unroll
        for (int i = 0; i < S::RTW; ++i) xz[i] = ld4(s_z + ((wr * S::RTW + i) * 16 + r) * LDZ + kb + 4 * q);
#pragma unroll
        for (int j = 0; j < S::HCW; ++j) {
            const int col = (wc * S::HCW + j) * 16 + r;
            const float* wp = Wvc_g + (int64_t)col * 2 * H + kb + 4 * q;
            const float4 w_r = ld4(wp), w_z = ld4(wp + 2 * H * H), w_n = ld4(wp + 4 * H * H);
#pragma unroll
            for (int i = 0; i < S::RTW; ++i) {
                mma_kblock(ar[i][j], xz[i], w_r);
                mma_kblock(az[i][j], xz[i], w_z);
                mma_kblock(an[i][j], xz[i], w_n);
            }
        }
    }
}

template <int H, bool HID = false>
__global__ __launch_bounds__(kThreads) void k_level_fwd(LevelArgs a) {
    using S = WaveSplit<H>;
    using M = LevelSmem<H, false>;
    constexpr int LDZ = M::LDZ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_z = smem + M::off_z;
    float* s_o = smem + M::off_z;      // reused once the MFMAs have consumed zbar
    float* s_sa = smem + M::off_sa;
    int* s_node = reinterpret_cast<int*>(smem + M::off_node);
    const int tile = a.tile_begin + blockIdx.x;
    const int start = a.tile_start[tile], count = a.tile_count[tile], g = a.tile_slot[tile];
    level_stage_vectors<H, false>(a, g, smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
    const int grp = tid / S::LPR, lr = tid % S::LPR;
    __syncthreads();
    const float4 us = ld4(smem + M::off_u + 4 * lr), uf = ld4(smem + M::off_u + H + 4 * lr);
    for (int row = grp; row < kTileRows; row += S::GROUPS) {
        float4 zs = zero4(), zf = zero4();
        float sa = 0.f;
        int node = -1;
        if (row < count) {
            node = a.order[start + row];
            float m, inv;
            attn_row<H>(a, node, us, uf, lr, m, inv, zs, zf);
            sa = a.in_ptr[node + 1] > a.in_ptr[node] ? 1.0f : 0.0f;
        }
        st4(s_z + row * LDZ + 4 * lr, zs);
        st4(s_z + row * LDZ + H + 4 * lr, zf);
        if (lr == 0) { s_sa[row] = sa; s_node[row] = node; }
    }
    __syncthreads();
    f32x4 ar[S::RTW][S::HCW], az[S::RTW][S::HCW], an[S::RTW][S::HCW];
    level_gemm<H>(a.Wvc + (int64_t)g * 3 * H * 2 * H, s_z, ar, az, an);
    __syncthreads();                   // s_o overlays s_z
    const float* s_bvc = smem + M::off_bvc; const float* s_bih = smem + M::off_bih; const float* s_bhh = smem + M::off_bhh;
#pragma unroll
    for (int i = 0; i < S::RTW; ++i)
#pragma unroll
        for (int j = 0; j < S::HCW; ++j) {
            const int col = (wc * S::HCW + j) * 16 + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                const float sa = s_sa[row];
                float gr = 0.f, gz = 0.f, gn = 0.f, hp = 0.f;
                if constexpr (HID) {       // previous-round state: gh = W_hh h + b_hh per gate block, h' = (1 - z) n + z h
                    const int64_t node = s_node[row];
                    if (node >= 0) { const float* g_ = a.gh + node * 3 * H + col; gr = g_[0]; gz = g_[H]; gn = g_[2 * H]; hp = a.hprev[node * H + col]; }
                }
                const float rr = sigmoidf_(ar[i][j][e] + sa * s_bvc[col] + s_bih[col] + s_bhh[col] + gr);
                const float zz = sigmoidf_(az[i][j][e] + sa * s_bvc[H + col] + s_bih[H + col] + s_bhh[H + col] + gz);
                const float nn = tanhf_(an[i][j][e] + sa * s_bvc[2 * H + col] + s_bih[2 * H + col] + rr * (s_bhh[2 * H + col] + gn));
                s_o[row * S::LD + col] = (1.0f - zz) * nn + zz * hp;        // round 1: h0 = 0
            }
        }
    __syncthreads();
    for (int row = grp; row < count; row += S::GROUPS)
        st4(a.hf + (int64_t)s_node[row] * H + 4 * lr, ld4(s_o + row * S::LD + 4 * lr));
}

// gradient a node's rows receive from its consumers' attention inputs
template <int H>
__device__ __forceinline__ void pull_row(const LevelArgs& a, int64_t node, int lr, float4& gs, float4& gf) {
    const int e0 = a.out_ptr[node], e1 = a.out_ptr[node + 1];
    gs = zero4(); gf = zero4();
    for (int e = e0; e < e1; ++e) {
        const int64_t c = a.out_dst[e];
        const int gc = a.gslot[c];
        if (gc == kNoGate) continue;
        const int sl = a.out_slot[e];
        const float al = a.alpha[sl], ds = a.dsc[sl];
        const float* dz = a.dzb + c * 2 * H;
        const float* u = a.attn_u + (int64_t)gc * 2 * H;
        gs = fma4(al, ld4(dz + 4 * lr), fma4(ds, ld4(u + 4 * lr), gs));
        gf = fma4(al, ld4(dz + H + 4 * lr), fma4(ds, ld4(u + H + 4 * lr), gf));
    }
}

template <int H>
__device__ __forceinline__ void colsum_lds(float v, float* dst) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if ((threadIdx.x & 63) < 16) atomicAdd(dst, v);
}

template <int H, bool HID = false>
__global__ __launch_bounds__(kThreads) void k_level_bwd(LevelArgs a) {
    using S = WaveSplit<H>;
    using S2 = WaveSplit<2 * H>;
    using M = LevelSmem<H>;
    constexpr int LDZ = M::LDZ;
    constexpr int TI = H / 16, TJ = 2 * H / 16, TT = TI * TJ;
    constexpr bool SPLITK = TT < 4;
    constexpr int TPW = SPLITK ? TT : TT / 4;
    constexpr int KSTEPS = SPLITK ? 4 : 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_z = smem + M::off_z;
    float* s_d = smem + M::off_o;      // dh tile, then the three dG tiles
    float* s_dz = smem + M::off_dz;
    float* s_sa = smem + M::off_sa;
    float* s_m = smem + M::off_m;
    float* s_inv = smem + M::off_inv;
    int* s_node = reinterpret_cast<int*>(smem + M::off_node);
    float* s_gu = smem + M::off_gu;
    float* s_dbvc = smem + M::off_dbvc; float* s_dbih = smem + M::off_dbih; float* s_dbhh = smem + M::off_dbhh;
    const int tile = a.tile_begin + blockIdx.x;
    const int start = a.tile_start[tile], count = a.tile_count[tile], g = a.tile_slot[tile];
    level_stage_vectors<H, true>(a, g, smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
    const int grp = tid / S::LPR, lr = tid % S::LPR;
    for (int i = tid; i < M::bwd_floats - M::off_gu; i += kThreads) smem[M::off_gu + i] = 0.f;
    __syncthreads();
    const float4 us = ld4(smem + M::off_u + 4 * lr), uf = ld4(smem + M::off_u + H + 4 * lr);
    // ---- 0/1. pull dL/dhf (and finish dL/dhs) of the tile's nodes, recompute their attention
    for (int row = grp; row < kTileRows; row += S::GROUPS) {
        float4 zs = zero4(), zf = zero4(), dh = zero4();
        float sa = 0.f, m = 0.f, inv = 0.f;
        int node = -1;
        if (row < count) {
            node = a.order[start + row];
            float4 gs, gf;
            pull_row<H>(a, node, lr, gs, gf);
            dh = add4(gf, ld4(a.ghf + (int64_t)node * H + 4 * lr));
            float* gp = a.ghs + (int64_t)node * H + 4 * lr;
            st4(gp, add4(ld4(gp), gs));
            attn_row<H>(a, node, us, uf, lr, m, inv, zs, zf);
            sa = a.in_ptr[node + 1] > a.in_ptr[node] ? 1.0f : 0.0f;
        }
        st4(s_z + row * LDZ + 4 * lr, zs);
        st4(s_z + row * LDZ + H + 4 * lr, zf);
        st4(s_d + row * S::LD + 4 * lr, dh);
        if (lr == 0) { s_sa[row] = sa; s_m[row] = m; s_inv[row] = inv; s_node[row] = node; }
    }
    __syncthreads();
    // ---- 2. recompute gates
    f32x4 ar[S::RTW][S::HCW], az[S::RTW][S::HCW], an[S::RTW][S::HCW];
    level_gemm<H>(a.Wvc + (int64_t)g * 3 * H * 2 * H, s_z, ar, az, an);
    const float* s_bvc = smem + M::off_bvc; const float* s_bih = smem + M::off_bih; const float* s_bhh = smem + M::off_bhh;
    // ---- 3. GRU backward (h0 = 0: hf = (1-z) n, gh = b_hh); ar/az/an become da_r/da_z/da_n
#pragma unroll
    for (int j = 0; j < S::HCW; ++j) {
        const int col = (wc * S::HCW + j) * 16 + r;
        const float bhn = s_bhh[2 * H + col];
        float b_r = 0.f, b_z = 0.f, b_n = 0.f, v_r = 0.f, v_z = 0.f, v_n = 0.f, h_n = 0.f;
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                const float sa = s_sa[row];
                float gr = 0.f, gz = 0.f, gn = 0.f, hp = 0.f;
                int64_t node = -1;
                if constexpr (HID) {
                    node = s_node[row];
                    if (node >= 0) { const float* g_ = a.gh + node * 3 * H + col; gr = g_[0]; gz = g_[H]; gn = g_[2 * H]; hp = a.hprev[node * H + col]; }
                }
                const float ghn = bhn + gn;
                const float rr = sigmoidf_(ar[i][j][e] + sa * s_bvc[col] + s_bih[col] + s_bhh[col] + gr);
                const float zz = sigmoidf_(az[i][j][e] + sa * s_bvc[H + col] + s_bih[H + col] + s_bhh[H + col] + gz);
                const float nn = tanhf_(an[i][j][e] + sa * s_bvc[2 * H + col] + s_bih[2 * H + col] + rr * ghn);
                const float dh = s_d[row * S::LD + col];
                const float dan = dh * (1.0f - zz) * (1.0f - nn * nn);
                const float daz = dh * (hp - nn) * zz * (1.0f - zz);
                const float dar = dan * ghn * rr * (1.0f - rr);
                if constexpr (HID) {
                    if (node >= 0) {
                        float* d_ = a.dgh + node * 3 * H + col;
                        d_[0] = dar; d_[H] = daz; d_[2 * H] = dan * rr;       // d(gh): the caller's linear kernels carry it to W_hh, b_hh, h_prev
                        a.ghprev[node * H + col] = dh * zz;                   // the direct path to the previous state
                    }
                }
                ar[i][j][e] = dar; az[i][j][e] = daz; an[i][j][e] = dan;
                b_r += dar; b_z += daz; b_n += dan; h_n += dan * rr;
                v_r += sa * dar; v_z += sa * daz; v_n += sa * dan;
            }
        colsum_lds<H>(b_r, s_dbih + col); colsum_lds<H>(b_z, s_dbih + H + col); colsum_lds<H>(b_n, s_dbih + 2 * H + col);
        colsum_lds<H>(b_r, s_dbhh + col); colsum_lds<H>(b_z, s_dbhh + H + col); colsum_lds<H>(h_n, s_dbhh + 2 * H + col);
        colsum_lds<H>(v_r, s_dbvc + col); colsum_lds<H>(v_z, s_dbvc + H + col); colsum_lds<H>(v_n, s_dbvc + 2 * H + col);
    }
    // ---- 4. three passes: d(zbar) += dG_p * Wvc[p]  and  dWvc[p] += dG_p^T * zbar
    const int wc2 = w % S2::WPC, wr2 = w / S2::WPC;
    f32x4 dz[S2::RTW][S2::HCW];
#pragma unroll
    for (int i = 0; i < S2::RTW; ++i)
#pragma unroll
        for (int j = 0; j < S2::HCW; ++j) dz[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* WvcT_g = a.WvcT + (int64_t)g * 2 * H * 3 * H;
    float* dWvc_g = a.dWvc + (int64_t)g * 3 * H * 2 * H;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        __syncthreads();               // readers of s_d (phase 3, or the previous pass)
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int j = 0; j < S::HCW; ++j) {
                const int col = (wc * S::HCW + j) * 16 + r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                    s_d[row * S::LD + col] = p == 0 ? ar[i][j][e] : p == 1 ? az[i][j][e] : an[i][j][e];
                }
            }
        __syncthreads();
#pragma unroll 1
        for (int kb = 0; kb < H; kb += 16) {
            float4 xd[S2::RTW];
#pragma unroll
            for (int i = 0; i < S2::RTW; ++i) xd[i] = ld4(s_d + ((wr2 * S2::RTW + i) * 16 + r) * S::LD + kb + 4 * q);
#pragma unroll
            for (int j = 0; j < S2::HCW; ++j) {
                const int col = (wc2 * S2::HCW + j) * 16 + r;
                const float4 wv = ld4(WvcT_g + (int64_t)col * 3 * H + p * H + kb + 4 * q);
#pragma unroll
                for (int i = 0; i < S2::RTW; ++i) mma_kblock(dz[i][j], xd[i], wv);
            }
        }
        // weight gradient of this gate block, flushed straight away (one workgroup = one tile)
        f32x4 gw[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) gw[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int row = (SPLITK ? 16 * w : 0) + kk * 4 + q;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tl = SPLITK ? t : w * TPW + t;
                const int it = tl / TJ, jt = tl % TJ;
                gw[t] = mfma16(s_d[row * S::LD + it * 16 + r], s_z[row * LDZ + jt * 16 + r], gw[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tl = SPLITK ? t : w * TPW + t;
            const int it = tl / TJ, jt = tl % TJ;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                atomicAdd(dWvc_g + (int64_t)(p * H + it * 16 + q * 4 + e) * 2 * H + jt * 16 + r, gw[t][e]);
        }
    }
    // ---- 5. d(zbar) tile to LDS (row layout needed for the attention backward); it overlays the dG tile
    __syncthreads();
#pragma unroll
    for (int i = 0; i < S2::RTW; ++i)
#pragma unroll
        for (int j = 0; j < S2::HCW; ++j) {
            const int col = (wc2 * S2::HCW + j) * 16 + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) s_dz[((wr2 * S2::RTW + i) * 16 + q * 4 + e) * LDZ + col] = dz[i][j][e];
        }
    __syncthreads();
    // ---- 6. attention backward per in-edge: leave alpha, d(score) for the sources' pulls, d(zbar) per node
    float4 gus = zero4(), guf = zero4();
    for (int row = grp; row < count; row += S::GROUPS) {
        const int64_t node = s_node[row];
        const float4 dzs = ld4(s_dz + row * LDZ + 4 * lr), dzf = ld4(s_dz + row * LDZ + H + 4 * lr);
        const float4 zs = ld4(s_z + row * LDZ + 4 * lr), zf = ld4(s_z + row * LDZ + H + 4 * lr);
        st4(a.dzb + node * 2 * H + 4 * lr, dzs);
        st4(a.dzb + node * 2 * H + H + 4 * lr, dzf);
        const float ci = group_sum<S::LPR>(dot4(dzs, zs) + dot4(dzf, zf));
        const float m = s_m[row], inv = s_inv[row];
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

// nodes the sweep never updates (primary inputs, unknown gate types): only their hs rows feed consumers
template <int H>
__global__ __launch_bounds__(kThreads) void k_level_pull_inactive(LevelArgs a) {
    constexpr int LPR = H / 4;
    const int lr = threadIdx.x % LPR;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / LPR);
    for (int64_t node = (int64_t)blockIdx.x * (kThreads / LPR) + threadIdx.x / LPR; node < a.N; node += stride) {
        if (a.gslot[node] != kNoGate) continue;
        float4 gs, gf;
        pull_row<H>(a, node, lr, gs, gf);
        float* gp = a.ghs + node * H + 4 * lr;
        st4(gp, add4(ld4(gp), gs));
    }
}

template <int H, bool HID>
int launch_level_v(bool bwd, const LevelArgs& a, int ntiles, hipStream_t st) {
    using M = LevelSmem<H>;
    const size_t shm = (bwd ? (size_t)M::bwd_floats : (size_t)LevelSmem<H, false>::fwd_floats) * sizeof(float);
    if (bwd) {
        static bool set_b = false;
        if (!set_b) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_level_bwd<H, HID>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set_b = true; }
        hipLaunchKernelGGL((k_level_bwd<H, HID>), dim3(ntiles), dim3(kThreads), shm, st, a);
    } else {
        static bool set_f = false;
        if (!set_f) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_level_fwd<H, HID>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set_f = true; }
        hipLaunchKernelGGL((k_level_fwd<H, HID>), dim3(ntiles), dim3(kThreads), shm, st, a);
    }
    MGV_LAUNCH_RET();
}

template <int H>
int launch_level(bool bwd, const LevelArgs& a, int ntiles, hipStream_t st) {
    return a.gh ? launch_level_v<H, true>(bwd, a, ntiles, st) : launch_level_v<H, false>(bwd, a, ntiles, st);
}

}  // namespace mgv

// Runs levels [1, L) forward.  level_tile_ptr is a HOST array of L+1 tile offsets.
static int sweep_fwd_impl(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                  const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                                  const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src, const float* hs,
                                  float* hf, const float* attn_u, const float* Wvc, const float* bvc, const float* bih,
                                  const float* bhh, const float* gh, const float* h_prev, void* stream) {
    MGV_CHECK_ARG(N >= 0 && T >= 1 && num_levels >= 0 && level_tile_ptr_host && hs && hf && attn_u && Wvc && bvc && bih && bhh && in_ptr);
    MGV_CHECK_ARG((gh == nullptr) == (h_prev == nullptr));
    mgv::LevelArgs a{};
    a.gh = gh; a.hprev = h_prev;
    a.N = N; a.T = T; a.order = order; a.tile_start = tile_start; a.tile_count = tile_count; a.tile_slot = tile_slot;
    a.in_ptr = in_ptr; a.in_src = in_src; a.hs = hs; a.hf = hf; a.attn_u = attn_u; a.Wvc = Wvc; a.bvc = bvc; a.bih = bih; a.bhh = bhh;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int lv = 1; lv < num_levels; ++lv) {
        const int t0 = level_tile_ptr_host[lv], t1 = level_tile_ptr_host[lv + 1];
        if (t1 <= t0) continue;
        MGV_CHECK_ARG(order && tile_start && tile_count && tile_slot && in_src);
        a.tile_begin = t0;
        int rc;
        switch (H) {
            case 16: rc = mgv::launch_level<16>(false, a, t1 - t0, st); break;
            case 32: rc = mgv::launch_level<32>(false, a, t1 - t0, st); break;
            case 64: rc = mgv::launch_level<64>(false, a, t1 - t0, st); break;
            default: return MGV_EUNSUPPORTED;
        }
        if (rc != MGV_OK) return rc;
    }
    return MGV_OK;
}

extern "C" int mgv_func_sweep_fwd(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                  const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                                  const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src, const float* hs,
                                  float* hf, const float* attn_u, const float* Wvc, const float* bvc, const float* bih,
                                  const float* bhh, void* stream) {
    return sweep_fwd_impl(H, N, T, num_levels, level_tile_ptr_host, order, tile_start, tile_count, tile_slot, in_ptr, in_src, hs, hf, attn_u, Wvc,
                          bvc, bih, bhh, nullptr, nullptr, stream);
}

extern "C" int mgv_func_sweep_round_fwd(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                        const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                                        const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src, const float* hs,
                                        float* hf, const float* attn_u, const float* Wvc, const float* bvc, const float* bih,
                                        const float* zero_bhh, const float* gh, const float* h_prev, void* stream) {
    MGV_CHECK_ARG(gh && h_prev);
    return sweep_fwd_impl(H, N, T, num_levels, level_tile_ptr_host, order, tile_start, tile_count, tile_slot, in_ptr, in_src, hs, hf, attn_u, Wvc,
                          bvc, bih, zero_bhh, gh, h_prev, stream);
}

static int sweep_bwd_impl(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                  const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                                  const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src,
                                  const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_slot,
                                  const uint8_t* gslot, const float* hs, const float* hf, const float* attn_u,
                                  const float* Wvc, const float* WvcT, const float* bvc, const float* bih, const float* bhh,
                                  const float* ghf, float* ghs, float* dzb, float* alpha, float* dsc, float* d_attn_u,
                                  float* dWvc, float* dbvc, float* dbih, float* dbhh, const float* gh, const float* h_prev,
                                  float* d_gh, float* g_hprev, void* stream) {
    MGV_CHECK_ARG(N >= 0 && T >= 1 && num_levels >= 0 && level_tile_ptr_host && hs && hf && attn_u && Wvc && WvcT && bvc && bih && bhh);
    MGV_CHECK_ARG(in_ptr && out_ptr && gslot && ghf && ghs && dzb && d_attn_u && dWvc && dbvc && dbih && dbhh);
    MGV_CHECK_ARG(gh == nullptr ? (!h_prev && !d_gh && !g_hprev) : (h_prev && d_gh && g_hprev));
    if (N == 0) return MGV_OK;
    mgv::LevelArgs a{};
    a.gh = gh; a.hprev = h_prev; a.dgh = d_gh; a.ghprev = g_hprev;
    a.N = N; a.T = T; a.order = order; a.tile_start = tile_start; a.tile_count = tile_count; a.tile_slot = tile_slot;
    a.in_ptr = in_ptr; a.in_src = in_src; a.hs = hs; a.hf = const_cast<float*>(hf); a.attn_u = attn_u; a.Wvc = Wvc; a.bvc = bvc;
    a.bih = bih; a.bhh = bhh; a.WvcT = WvcT; a.out_ptr = out_ptr; a.out_dst = out_dst; a.out_slot = out_slot; a.gslot = gslot;
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
            case 16: rc = mgv::launch_level<16>(true, a, t1 - t0, st); break;
            case 32: rc = mgv::launch_level<32>(true, a, t1 - t0, st); break;
            case 64: rc = mgv::launch_level<64>(true, a, t1 - t0, st); break;
            default: return MGV_EUNSUPPORTED;
        }
        if (rc != MGV_OK) return rc;
    }
    // hs rows of nodes the sweep never updates
    const int rows_per_block = mgv::kThreads / (H / 4);
    const int grid = mgv::grid_for((N + rows_per_block - 1) / rows_per_block, 8);
    switch (H) {
        case 16: hipLaunchKernelGGL(mgv::k_level_pull_inactive<16>, dim3(grid), dim3(mgv::kThreads), 0, st, a); break;
        case 32: hipLaunchKernelGGL(mgv::k_level_pull_inactive<32>, dim3(grid), dim3(mgv::kThreads), 0, st, a); break;
        case 64: hipLaunchKernelGGL(mgv::k_level_pull_inactive<64>, dim3(grid), dim3(mgv::kThreads), 0, st, a); break;
        default: return MGV_EUNSUPPORTED;
    }
    MGV_LAUNCH_RET();
}

extern "C" int mgv_func_sweep_bwd(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                  const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                                  const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src,
                                  const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_slot,
                                  const uint8_t* gslot, const float* hs, const float* hf, const float* attn_u,
                                  const float* Wvc, const float* WvcT, const float* bvc, const float* bih, const float* bhh,
                                  const float* ghf, float* ghs, float* dzb, float* alpha, float* dsc, float* d_attn_u,
                                  float* dWvc, float* dbvc, float* dbih, float* dbhh, void* stream) {
    return sweep_bwd_impl(H, N, T, num_levels, level_tile_ptr_host, order, tile_start, tile_count, tile_slot, in_ptr, in_src, out_ptr, out_dst,
                          out_slot, gslot, hs, hf, attn_u, Wvc, WvcT, bvc, bih, bhh, ghf, ghs, dzb, alpha, dsc, d_attn_u, dWvc, dbvc, dbih, dbhh,
                          nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int mgv_func_sweep_round_bwd(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                        const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                                        const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src,
                                        const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_slot,
                                        const uint8_t* gslot, const float* hs, const float* hf, const float* attn_u,
                                        const float* Wvc, const float* WvcT, const float* bvc, const float* bih, const float* zero_bhh,
                                        const float* ghf, float* ghs, float* dzb, float* alpha, float* dsc, float* d_attn_u,
                                        float* dWvc, float* dbvc, float* dbih, float* dbhh_unused, const float* gh, const float* h_prev,
                                        float* d_gh, float* g_hprev, void* stream) {
    MGV_CHECK_ARG(gh && h_prev && d_gh && g_hprev);
    return sweep_bwd_impl(H, N, T, num_levels, level_tile_ptr_host, order, tile_start, tile_count, tile_slot, in_ptr, in_src, out_ptr, out_dst,
                          out_slot, gslot, hs, hf, attn_u, Wvc, WvcT, bvc, bih, zero_bhh, ghf, ghs, dzb, alpha, dsc, d_attn_u, dWvc, dbvc, dbih,
                          dbhh_unused, gh, h_prev, d_gh, g_hprev, stream);
}
