// Device pieces shared by the per-level sweep kernels (func_level_x3.hip) and the persistent sweep kernels
// (sweep_persist_x3.hip): argument block, LDS carving, index staging, attention forward / backward rows, the gate product.
#pragma once
#include "mgv_x3.h"
#include "mgv_stamps.h"
#include "mgv_slab.h"
#include "mgv_gridbar.h"
#include <algorithm>
#include "../../include/mgvae_hip.h"

namespace mgv {

constexpr uint8_t kNoGateX = 255;
constexpr int kLW = 8;                 // waves per workgroup
constexpr int kLT = 64 * kLW;          // threads per workgroup
constexpr int kInCap = 4;              // in-edges per row staged in LDS
constexpr int kInRegs = 3;             // ... of which this many source rows are gathered together (longer lists continue one by one)
constexpr int kOutCap = 16;            // out-edges per row staged in LDS
constexpr int kOutChunk = 2;           // consumer rows in flight per lane while pulling
constexpr int kWgradGrid = 256;         // workgroups of the deferred weight-gradient kernel (one slab row each)
constexpr int kRowInts = 32;            // packed order rows: 128 bytes per updated node
constexpr int kRowOut = 8;              // consumers per packed row
constexpr int kMaxSlots = 6;           // gate types whose attention vectors are kept in LDS

struct LevelX3Args {
    unsigned long long* stamps;   // diagnostic build only (MGV_STAMPS)
    int64_t N;
    int T;
    const int32_t* order; const int32_t* order_span; const int32_t* tile_start; const int32_t* tile_count; const int32_t* tile_slot;
    // ints per row of order_span: 4 = {in0, in1, out0, out1}; kRowInts = packed rows (GraphPlan.order_rows): the spans, then the first
    // 4 in-edge sources, the first kRowOut consumers as (node, in-CSR slot) pairs and their gate slots — a tile then reaches its
    // rows' lists with ONE load level instead of span -> edge lists -> gslot
    int span_ints;
    int tile_begin;
    const int32_t* in_ptr; const int32_t* in_src;
    const float* hs; float* hf;
    const float* attn_u;   // [T][2H]
    const __bf16* wpack;   // [T][4][6H^2]
    const float* bvc; const float* bih; const float* bhh;   // [T][3H]
    const int32_t* out_ptr; const int32_t* out_dst; const int32_t* out_slot; const uint8_t* gslot;
    const float* ghf; float* ghs; float* dzb; float* alpha; float* dsc;
    float* d_attn_u; float* dWvc; float* dbvc; float* dbih; float* dbhh;
    float* wslab;          // [tiles of the widest level][T][11H]: per-workgroup sums of gu, dbvc, dbih, dbhh
    float* dgrows;         // [n_active][3H] gate gradients and
    float* zrows;          // [n_active][2H] zbar rows in sweep order, for the weight-gradient kernel
    const int32_t* slot_tiles;   // tiles grouped by slot
    int skip_inactive;           // > 0: the inactive-node pull skips nodes with more consumers (mgv_sweep_pull_heavy writes their rows)
    // updated gates with more than skip_active consumers (an inverter of a clock-like input): their pull comes from a per-level
    // pre-pass (k_pull_heavy_seg<H, true>): heavy_nodes[heavy_k0 .. heavy_k1) = this level's, ascending ids; heavy_pull[k][2H]
    int skip_active; int heavy_k0; int heavy_k1; const int32_t* heavy_nodes; const float* heavy_pull;
    // rounds >= 2 of the sweep (dg_ae_model_aig.py:70-97: the GRU of a node starts from the node's state of the previous round):
    // gh[node][3H] = W_hh h_prev + b_hh of the node's own aggregator (formed by the linear kernels before the sweep, the caller
    // passes bhh = 0 here), hprev[node][H]; the backward leaves d(gh)[node][3H] and d(h_prev)[node][H] = dh * z.  NULL in round 1.
    const float* gh; const float* hprev; float* dgh; float* ghprev;
};

// kLW waves over a (64 rows) x COLS output: across column tiles first, then row tiles
template <int COLS>
struct SplitL {
    static constexpr int CT = COLS / 16;
    static constexpr int WPC = CT < kLW ? CT : kLW;
    static constexpr int WPR = kLW / WPC;
    static_assert(WPR <= 4, "more waves than 16-row tiles");
    static constexpr int RTW = 4 / WPR;
    static constexpr int HCW = CT / WPC;
    static_assert(HCW == 1, "one column tile per wave");
};

template <int H>
struct LvlSmem {
    static constexpr int LPR = H / 4;                         // lanes per row
    static constexpr int GROUPS = kLT / LPR;                  // rows in flight
    static constexpr int RPG = kTileRows / GROUPS;            // rows per lane group
    static_assert(RPG >= 1 && RPG * GROUPS == kTileRows, "row groups must tile the 64 rows");
    static constexpr int LDO = H + 4;                         // fp32 output / dh tile row
    static constexpr int LDZP = 2 * H + 8;                    // bf16 elements per zbar plane row
    static constexpr int LDGP = H + 8;                        // bf16 elements per gate-gradient plane row
    static constexpr int LDZF = 2 * H + 4;                    // fp32 d(zbar) tile row
    static constexpr int ZPB = kTileRows * LDZP * 2;          // bytes of one zbar plane
    static constexpr int GPB = kTileRows * LDGP * 2;
    static constexpr int DHB = kTileRows * LDO * 4;
    static constexpr int o_zhi = 0;
    static constexpr int o_zlo = o_zhi + ZPB;
    static_assert(DHB <= 2 * ZPB, "forward output tile reuses the zbar planes");
    static constexpr int SMALL_F = kMaxSlots * 2 * H + 9 * H + 3 * kTileRows;   // u of every slot, bvc/bih/bhh, sa/m/inv
    static constexpr int IDX_B = kTileRows * 4 + kTileRows * 16 + kTileRows * kInCap * 4;   // node, span, in-edge sources
    // forward
    static constexpr int o_small_f = o_zlo + ZPB;
    static constexpr int o_idx_f = o_small_f + SMALL_F * 4;
    static constexpr int fwd_bytes = o_idx_f + IDX_B;
    // backward: region R after the planes = dh fp32 | staged out-edges, then the gate-gradient planes, then d(zbar) fp32
    static constexpr int OUT_B = kTileRows * kOutCap * (4 + 4 + 4 + 1);   // consumer, alpha, d score, consumer slot
    static constexpr int o_r = o_zlo + ZPB;
    static constexpr int o_out = o_r + DHB;
    static constexpr int R_BYTES = (kTileRows * LDZF * 4 > DHB + OUT_B) ? kTileRows * LDZF * 4 : DHB + OUT_B;
    static_assert(2 * GPB <= R_BYTES, "region R");
    static constexpr int o_small_b = o_r + R_BYTES;
    static constexpr int o_acc = o_small_b + SMALL_F * 4;     // gu[2H], dbvc, dbih, dbhh [3H each]
    static constexpr int o_idx_b = o_acc + (2 * H + 9 * H) * 4;
    static constexpr int bwd_bytes = o_idx_b + IDX_B;
    static_assert(bwd_bytes <= 80 * 1024, "two backward workgroups per CU");
};

struct LvlSmall { float* uall; float* u; float* bvc; float* bih; float* bhh; float* sa; float* m; float* inv; };
struct LvlIdx { int* node; int4* span; int* insrc; };

template <int H>
__device__ __forceinline__ LvlSmall lvl_small(const LevelX3Args& a, int g, float* base) {
    LvlSmall v;
    v.uall = base; v.u = base + g * 2 * H; v.bvc = base + kMaxSlots * 2 * H; v.bih = v.bvc + 3 * H; v.bhh = v.bih + 3 * H;
    v.sa = v.bhh + 3 * H; v.m = v.sa + kTileRows; v.inv = v.m + kTileRows;
    for (int i = threadIdx.x; i < a.T * 2 * H; i += kLT) v.uall[i] = a.attn_u[i];
    for (int i = threadIdx.x; i < 3 * H; i += kLT) {
        v.bvc[i] = a.bvc[(int64_t)g * 3 * H + i]; v.bih[i] = a.bih[(int64_t)g * 3 * H + i]; v.bhh[i] = a.bhh[(int64_t)g * 3 * H + i];
    }
    return v;
}

__device__ __forceinline__ LvlIdx lvl_idx(unsigned char* base) {
    LvlIdx x;
    x.span = reinterpret_cast<int4*>(base);
    x.node = reinterpret_cast<int*>(base + kTileRows * 16);
    x.insrc = x.node + kTileRows;
    return x;
}

// rows' node ids and CSR spans {in0, in1, out0, out1}; padding rows get node -1 and empty spans.  Packed rows: the in-edge sources too.
__device__ __forceinline__ void stage_spans(const LevelX3Args& a, int start, int count, const LvlIdx& x) {
    const int stride = a.span_ints;
    if (threadIdx.x < kTileRows) {
        const int row = threadIdx.x;
        int node = -1;
        int4 sp = make_int4(0, 0, 0, 0);
        if (row < count) {
            node = a.order[start + row];
            sp = *reinterpret_cast<const int4*>(a.order_span + (int64_t)stride * (start + row));
        }
        x.node[row] = node;
        x.span[row] = sp;
    }
    if (stride == kRowInts && threadIdx.x < kTileRows * kInCap) {
        const int row = threadIdx.x / kInCap, k = threadIdx.x % kInCap;
        if (row < count) x.insrc[threadIdx.x] = a.order_span[(int64_t)stride * (start + row) + 4 + k];
    }
}

__device__ __forceinline__ void stage_in_edges(const LevelX3Args& a, const LvlIdx& x) {
    if (threadIdx.x < kTileRows * kInCap) {
        const int row = threadIdx.x / kInCap, k = threadIdx.x % kInCap;
        const int4 sp = x.span[row];
        if (sp.x + k < sp.y) x.insrc[threadIdx.x] = a.in_src[sp.x + k];
    }
}

struct OutStage { int* c; float* al; float* ds; uint8_t* gc; };

__device__ __forceinline__ OutStage out_stage(unsigned char* base) {
    OutStage o;
    o.c = reinterpret_cast<int*>(base);
    o.al = reinterpret_cast<float*>(base + kTileRows * kOutCap * 4);
    o.ds = reinterpret_cast<float*>(base + kTileRows * kOutCap * 8);
    o.gc = reinterpret_cast<uint8_t*>(base + kTileRows * kOutCap * 12);
    return o;
}

__device__ __forceinline__ void stage_out_edges(const LevelX3Args& a, const LvlIdx& x, const OutStage& o) {
    for (int i = threadIdx.x; i < kTileRows * kOutCap; i += kLT) {
        const int row = i / kOutCap, k = i % kOutCap;
        const int4 sp = x.span[row];
        if (sp.z + k < sp.w) {
            const int c = a.out_dst[sp.z + k], sl = a.out_slot[sp.z + k];
            const uint8_t gc = a.gslot[c];
            float al = 0.f, ds = 0.f;
            if (gc != kNoGateX) { al = a.alpha[sl]; ds = a.dsc[sl]; }
            o.c[i] = c; o.al[i] = al; o.ds[i] = ds; o.gc[i] = gc;
        }
    }
}

// the same from the packed rows: one thread per (row, consumer k < kRowOut); no dependence on the spans in LDS
__device__ __forceinline__ void stage_out_edges_packed(const LevelX3Args& a, int start, int count, const OutStage& o) {
    const int row = threadIdx.x / kRowOut, k = threadIdx.x % kRowOut;
    if (row < count) {
        const int32_t* rp = a.order_span + (int64_t)kRowInts * (start + row);
        const int2 cs = *reinterpret_cast<const int2*>(rp + 8 + 2 * k);
        const uint8_t gc = reinterpret_cast<const uint8_t*>(rp + 24)[k];
        if (cs.x >= 0) {
            float al = 0.f, ds = 0.f;
            if (gc != kNoGateX) { al = a.alpha[cs.y]; ds = a.dsc[cs.y]; }
            const int i = row * kOutCap + k;
            o.c[i] = cs.x; o.al[i] = al; o.ds[i] = ds; o.gc[i] = gc;
        }
    }
}

__device__ __forceinline__ float4 f4(const f32x4& v) { return make_float4(v[0], v[1], v[2], v[3]); }

// source rows of a node's first kInRegs in-edges, loaded together into registers the untaken path never writes
template <int H>
struct InRows {
    f32x4 xs[kInRegs], xf[kInRegs];
    __device__ __forceinline__ void issue(const LevelX3Args& a, const int* insrc_row, int deg, int lr) {
#pragma unroll
        for (int k = 0; k < kInRegs; ++k)
            if (k < deg) {
                const int64_t j = insrc_row[k];
                xs[k] = *reinterpret_cast<const f32x4*>(a.hs + j * H + 4 * lr);
                xf[k] = *reinterpret_cast<const f32x4*>(a.hf + j * H + 4 * lr);
            }
    }
};

// softmax attention over the in-edges (online form, same operation order as func_level.hip)
template <int H>
__device__ __forceinline__ void attn_reduce(const LevelX3Args& a, const InRows<H>& L, const int4& sp, const float4& us,
                                            const float4& uf, int lr, float& m, float& inv, float4& zs, float4& zf) {
    constexpr int LPR = H / 4;
    const int deg = sp.y - sp.x;
    m = -INFINITY;
    float S = 0.f;
    zs = zero4(); zf = zero4();
#pragma unroll
    for (int k = 0; k < kInRegs; ++k)
        if (k < deg) {
            const float4 xs = f4(L.xs[k]), xf = f4(L.xf[k]);
            const float sc = group_sum<LPR>(dot4(us, xs) + dot4(uf, xf));
            const float mn = fmaxf(m, sc);
            const float corr = __expf(m - mn), w = __expf(sc - mn);
            S = S * corr + w;
            zs = fma4(w, xs, scale4(corr, zs));
            zf = fma4(w, xf, scale4(corr, zf));
            m = mn;
        }
    for (int e = sp.x + kInRegs; e < sp.y; ++e) {
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
    if (deg <= 0) m = 0.f;
}

template <int H>
__device__ __forceinline__ void store_zbar(__bf16* z_hi, __bf16* z_lo, int row, int lr, const float4& zs, const float4& zf) {
    constexpr int LDZP = 2 * H + 8;
    bf16x4 hi, lo;
    split4(zs, hi, lo);
    st_bf4(z_hi + row * LDZP + 4 * lr, hi); st_bf4(z_lo + row * LDZP + 4 * lr, lo);
    split4(zf, hi, lo);
    st_bf4(z_hi + row * LDZP + H + 4 * lr, hi); st_bf4(z_lo + row * LDZP + H + 4 * lr, lo);
}

// gate pre-activations (r, z, n blocks) = zbar[64 x 2H] * Wvc_g^T from the split planes.  KU k-steps' weight fragments are requested
// together (one L2 round trip per KU k-steps: the forward kernel, which has the registers, asks for two at a time; KU = 1 elsewhere)
template <int H, int KU_ = 1>
__device__ __forceinline__ void lvl_gemm_x3(const __bf16* wslot, const __bf16* z_hi, const __bf16* z_lo,
                                            f32x4 (&ar)[SplitL<H>::RTW], f32x4 (&az)[SplitL<H>::RTW], f32x4 (&an)[SplitL<H>::RTW]) {
    using S = SplitL<H>;
    constexpr int LDZP = 2 * H + 8, BLK = 6 * H * H, KS = 2 * H / 32;
    constexpr int KU = (KS % KU_ == 0) ? KU_ : 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
#pragma unroll
    for (int i = 0; i < S::RTW; ++i) { ar[i] = f32x4{0.f, 0.f, 0.f, 0.f}; az[i] = ar[i]; an[i] = ar[i]; }
#pragma unroll 1
    for (int k0 = 0; k0 < KS; k0 += KU) {
        bf16x8 bh[KU][3], bl[KU][3];
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int wo = ((g * (H / 16) + wc) * KS + k0 + u) * 512 + lane * 8;
                bh[u][g] = ldfrag(wslot + wo); bl[u][g] = ldfrag(wslot + BLK + wo);
            }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int ks = k0 + u;
            bf16x8 xh[S::RTW], xl[S::RTW];
#pragma unroll
            for (int i = 0; i < S::RTW; ++i) {
                const int off = ((wr * S::RTW + i) * 16 + r) * LDZP + 32 * ks + 8 * q;
                xh[i] = ldfrag(z_hi + off); xl[i] = ldfrag(z_lo + off);
            }
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int i = 0; i < S::RTW; ++i) {
                    if (g == 0) mma_x3(ar[i], xh[i], xl[i], bh[u][g], bl[u][g]);
                    if (g == 1) mma_x3(az[i], xh[i], xl[i], bh[u][g], bl[u][g]);
                    if (g == 2) mma_x3(an[i], xh[i], xl[i], bh[u][g], bl[u][g]);
                }
        }
    }
}

__device__ __forceinline__ void colsum_lds_lx(float v, float* dst) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if ((threadIdx.x & 63) < 16) atomicAdd(dst, v);
}

// gradient a node's rows receive from its consumers' attention inputs, consumers beyond the staged list
template <int H>
__device__ __forceinline__ void pull_tail(const LevelX3Args& a, int e0, int e1, int lr, float4& gs, float4& gf) {
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

// consumers [k0, k0 + kOutChunk) of the staged list: their d(zbar) rows loaded together
template <int H>
struct OutRows {
    f32x4 ds_[kOutChunk], df_[kOutChunk];
    __device__ __forceinline__ void issue(const LevelX3Args& a, const OutStage& o, int base, int k0, int n, int lr) {
#pragma unroll
        for (int k = 0; k < kOutChunk; ++k)
            if (k0 + k < n && o.gc[base + k0 + k] != kNoGateX) {
                const float* dz = a.dzb + (int64_t)o.c[base + k0 + k] * 2 * H;
                ds_[k] = *reinterpret_cast<const f32x4*>(dz + 4 * lr);
                df_[k] = *reinterpret_cast<const f32x4*>(dz + H + 4 * lr);
            }
    }
    __device__ __forceinline__ void reduce(const OutStage& o, const float* uall, int base, int k0, int n, int lr, float4& gs, float4& gf) const {
#pragma unroll
        for (int k = 0; k < kOutChunk; ++k)
            if (k0 + k < n && o.gc[base + k0 + k] != kNoGateX) {
                const float al = o.al[base + k0 + k], ds = o.ds[base + k0 + k];
                const float* u = uall + (int)o.gc[base + k0 + k] * 2 * H;
                gs = fma4(al, f4(ds_[k]), fma4(ds, ld4(u + 4 * lr), gs));
                gf = fma4(al, f4(df_[k]), fma4(ds, ld4(u + H + 4 * lr), gf));
            }
    }
};

// attention backward of one row from its staged source rows: alpha and d(score) of the first kInRegs in-edges come
// back in registers (the caller stores them after it has issued the next row's loads); longer lists finish here.
// d(score_j) = alpha_j (t_j - sum_k alpha_k t_k), t_k = d(zbar) . x_k, is formed in the CENTRED way, alpha_j d(zbar) . (x_j - zbar)
// with zbar = sum_k alpha_k x_k re-formed here in fp32 from the source rows: the difference of two nearly equal rows is taken
// exactly before the dot product.  (Subtracting d(zbar) . zbar with zbar read back from its bf16 hi/lo planes — 2^-17 relative —
// left the attention-logit parameters attn_lin / msg_k with 1e-3 of their scale in error: their gradient IS this cancelling sum.)
// (WT: the per-edge scalars of the list's tail leave write-through, for the persistent sweep whose other workgroups read them in-launch)
template <int H, bool WT = false>
__device__ __forceinline__ void attn_bwd_row(const LevelX3Args& a, const InRows<H>& L, const int4& sp, const float4& us,
                                             const float4& uf, const float4& dzs, const float4& dzf, float m,
                                             float inv, int lr, float (&al)[kInRegs], float (&ds)[kInRegs], float4& gus, float4& guf) {
    constexpr int LPR = H / 4;
    const int deg = sp.y - sp.x;
    float4 zs = zero4(), zf = zero4();
#pragma unroll
    for (int k = 0; k < kInRegs; ++k)
        if (k < deg) {
            const float4 xs = f4(L.xs[k]), xf = f4(L.xf[k]);
            const float sc = group_sum<LPR>(dot4(us, xs) + dot4(uf, xf));
            al[k] = __expf(sc - m) * inv;
            zs = fma4(al[k], xs, zs); zf = fma4(al[k], xf, zf);
        }
    for (int e = sp.x + kInRegs; e < sp.y; ++e) {          // lists longer than the staged ones (no gate type of the reference has them)
        const int64_t j = a.in_src[e];
        const float4 xs = ld4(a.hs + j * H + 4 * lr), xf = ld4(a.hf + j * H + 4 * lr);
        const float al_e = __expf(group_sum<LPR>(dot4(us, xs) + dot4(uf, xf)) - m) * inv;
        zs = fma4(al_e, xs, zs); zf = fma4(al_e, xf, zf);
    }
#pragma unroll
    for (int k = 0; k < kInRegs; ++k)
        if (k < deg) {
            const float4 xs = f4(L.xs[k]), xf = f4(L.xf[k]);
            const float4 cs = make_float4(xs.x - zs.x, xs.y - zs.y, xs.z - zs.z, xs.w - zs.w), cf = make_float4(xf.x - zf.x, xf.y - zf.y, xf.z - zf.z, xf.w - zf.w);
            ds[k] = al[k] * group_sum<LPR>(dot4(dzs, cs) + dot4(dzf, cf));
            gus = fma4(ds[k], xs, gus);
            guf = fma4(ds[k], xf, guf);
        }
    for (int e = sp.x + kInRegs; e < sp.y; ++e) {
        const int64_t j = a.in_src[e];
        const float4 xs = ld4(a.hs + j * H + 4 * lr), xf = ld4(a.hf + j * H + 4 * lr);
        const float al_e = __expf(group_sum<LPR>(dot4(us, xs) + dot4(uf, xf)) - m) * inv;
        const float4 cs = make_float4(xs.x - zs.x, xs.y - zs.y, xs.z - zs.z, xs.w - zs.w), cf = make_float4(xf.x - zf.x, xf.y - zf.y, xf.z - zf.z, xf.w - zf.w);
        const float ds_e = al_e * group_sum<LPR>(dot4(dzs, cs) + dot4(dzf, cf));
        if (lr == 0) {
            if (WT) { st1_wt(a.alpha + e, al_e); st1_wt(a.dsc + e, ds_e); }
            else { a.alpha[e] = al_e; a.dsc[e] = ds_e; }
        }
        gus = fma4(ds_e, xs, gus);
        guf = fma4(ds_e, xf, guf);
    }
}

// nodes the sweep never updates (primary inputs, unknown gate types): only their hs rows feed consumers
template <int H>
__global__ __launch_bounds__(kThreads) void k_level_pull_inactive_x3(LevelX3Args a) {
    constexpr int LPR = H / 4;
    const int lr = threadIdx.x % LPR;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / LPR);
    for (int64_t node = (int64_t)blockIdx.x * (kThreads / LPR) + threadIdx.x / LPR; node < a.N; node += stride) {
        if (a.gslot[node] != kNoGateX) continue;
        if (a.skip_inactive > 0 && a.out_ptr[node + 1] - a.out_ptr[node] > a.skip_inactive) continue;      // left to mgv_sweep_pull_heavy
        float4 gs = zero4(), gf = zero4();
        // four consumers in flight: a primary input is where the long consumer lists are (a clock- or reset-like net)
        const int e1 = a.out_ptr[node + 1];
        int e = a.out_ptr[node];
        for (; e + 4 <= e1; e += 4) {
            int64_t c[4]; int gc[4], sl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { c[k] = a.out_dst[e + k]; sl[k] = a.out_slot[e + k]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) gc[k] = a.gslot[c[k]];
            float al[4], ds[4]; float4 dz[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                al[k] = 0.f; ds[k] = 0.f; dz[k] = zero4();
                if (gc[k] != kNoGateX) { al[k] = a.alpha[sl[k]]; ds[k] = a.dsc[sl[k]]; dz[k] = ld4(a.dzb + c[k] * 2 * H + 4 * lr); }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (gc[k] != kNoGateX) gs = fma4(al[k], dz[k], fma4(ds[k], ld4(a.attn_u + (int64_t)gc[k] * 2 * H + 4 * lr), gs));
        }
        pull_tail<H>(a, e, e1, lr, gs, gf);
        st4(a.ghs + node * H + 4 * lr, gs);
    }
}

}  // namespace mgv
