// Pieces shared by the bf16x3 struct-stage kernels (struct_stage_x3.hip: forward + first backward; struct_stage_bwd2_x3.hip:
// the register-resident-weight backward): kernel arguments, LDS carve-up of the first kernels, the persistent tile
// sequence, the neighbour-index prefetch, the row phase and the 2x2-blocked transposed-read weight gradient.
#pragma once
#include "mgv_x3.h"
#include "../../include/mgvae_hip.h"

namespace mgv {

struct StageX3Args {
    unsigned long long* stamps;   // diagnostic build only (MGV_STAMPS): [8 waves][16 phases] cycle sums
    int xcd;                      // 1: XCD-contiguous tile order (default), 0: round-robin (MGV_XCD_TILES=0, A/B measurements)
    int prefetch;                 // bwd2: 1: L2 prefetch of the next tile's rows (default), 0: off (MGV_ROW_PREFETCH=0, A/B measurements)
    int64_t N;
    const float* h_in;
    const int32_t* ptr;
    const int32_t* idx;
    const uint8_t* xcls;
    const float* xtab;
    int C;
    const __bf16* wpack;
    const float* bc;
    const float* bhh;
    const float* lnw;
    const float* lnb;
    float eps;
    float* h_out;
    const float* gy_direct;
    const float* gy_agg;
    float* g_direct_out;
    float* g_agg_out;
    float* dWc; float* dbc; float* dWhh; float* dbhh; float* dxtab; float* dlnw; float* dlnb;
    // rows with more than kHeavyRow neighbours (a clock/reset-like net), ascending node ids, and their neighbour sums formed by
    // k_heavy_sums before the stage kernel starts: [heavy_n][H] over h_in, then [heavy_n][H] over gy_agg (backward only)
    int heavy_n; const int32_t* heavy_nodes; const float* heavy_a; const float* heavy_g;
    // table mode (the half round after the (degree, class)-table one): h_in is the C-row table, a neighbour entry carries its class in
    // the top byte (h row = entry >> 24, gradient row = entry & 0xffffff), a node's own h row is own_idx[node].  Normal mode: 0 / ~0 / NULL.
    int hshift; int gmask; const int32_t* own_idx;
    // LayerNorm statistics of the stage's rows, [N][2] = {mean, rstd} of the pre-LayerNorm state: written by the forward kernel, read by
    // the bwd2 kernel (which then skips two of its four cross-lane row sums and the four-way combination); NULL: not kept / recomputed
    float* ln_stats;
};

#include "mgv_stamps.h"

constexpr int kMaxClsX3 = 8;
constexpr int kTPR = 2;                 // gate-gradient tiles staged in LDS per round of the backward phase E (4 spills registers: 7.4 vs 4.6 ms)
constexpr int XLD = 24;                 // row-major [64][16 (+8 pad)] planes of [deg, onehot(cls) x8, 1, 0..]
constexpr int kNW = 8;                  // waves per workgroup (two per SIMD)
constexpr int kThreadsX3 = kNW * 64;
constexpr int kIdxCap = 512;            // neighbour entries of one tile kept in LDS (mean tile: ~100-140); larger lists take the generic path
constexpr int kPtrPad = 80;             // 65 CSR pointers of a tile + the tile's maximum degree at [72], padded

// 8 waves over a (64 rows) x (H hidden columns) output: across column tiles first, then row tiles
template <int H>
struct SplitX3 {
    static constexpr int HC = H / 16;
    static constexpr int WPC = HC < 4 ? HC : 4;
    static constexpr int WPR = kNW / WPC;
    static constexpr int RTW = 4 / WPR;
    static_assert(RTW >= 1, "too many waves for the tile");
    static constexpr int HCW = HC / WPC;
    static constexpr int LD = H + 4;
    static constexpr int LPR = H / 4;
    static constexpr int GROUPS = kThreadsX3 / LPR;
};

template <int H>
struct X3Smem {
    using S = SplitX3<H>;
    static constexpr int LDP = H + 8;                         // bf16 elements per row-major plane row
    static constexpr int PB = kTileRows * LDP * 2;            // bytes of a row-major plane
    static constexpr int F32TILE = kTileRows * S::LD * 4;
    static constexpr int SMALL_F = kMaxClsX3 * 3 * H + 3 * H + 3 * H + H + H + kTileRows + kTileRows;   // floats
    // forward
    static constexpr int f_planes = 0;                        // agg_hi, agg_lo, hin_hi, hin_lo
    static constexpr int f_hin = f_planes + 4 * PB;           // fp32 own rows
    static constexpr int f_pre = f_planes;                    // fp32 pre-LayerNorm rows: reuse the agg planes once the MFMAs are done
    static_assert(F32TILE <= 2 * PB, "pre-LN tile must fit the two agg planes");
    static constexpr int f_small = f_hin + F32TILE;
    static constexpr int IDX_BYTES = 2 * (kPtrPad + kIdxCap + 8) * 4;
    static constexpr int f_idx = f_small + SMALL_F * 4;
    static constexpr int fwd_bytes = f_idx + IDX_BYTES;
    // backward
    static constexpr int b_planes = 0;                        // region A: row-major operand planes
    static constexpr int b_c = b_planes + 4 * PB;             // region C: {pre, dy fp32} then {d_hi, d_lo}
    static constexpr int C_BYTES = (2 * F32TILE > 2 * kTPR * PB) ? 2 * F32TILE : 2 * kTPR * PB;   // kTPR gate-gradient tiles (hi, lo each) per round
    static constexpr int b_small = b_c + C_BYTES;
    static constexpr int b_stat = b_small + SMALL_F * 4;      // 4 floats per row
    static constexpr int b_acc = b_stat + 4 * kTileRows * 4;  // dxt[C*3H], dbc[3H], dbhh[3H], dlnw[H], dlnb[H]
    static constexpr int ACC_F = kMaxClsX3 * 3 * H + 3 * H + 3 * H + H + H;
    static constexpr int b_idx = b_acc + ACC_F * 4;
    static constexpr int b_xe = b_idx + IDX_BYTES;                 // xe_hi, xe_lo: [64][XLD] columns = deg, onehot(cls) x8, 1, 0...
    // hi planes of Wc and Whh, row-major [3H][H+8] bf16, resident for the whole kernel: the recompute reads them as plain
    // fragments, the dgrad reads the SAME rows transposed; only the lo planes still stream from L2 (half the L1 traffic)
    static constexpr int WLD = H + 8;
    static constexpr int WPB = 3 * H * WLD * 2;
    static constexpr int b_w = b_xe + 2 * kTileRows * XLD * 2;
    static constexpr int bwd_bytes = b_w + 2 * WPB;
    static_assert(bwd_bytes <= 160 * 1024, "backward LDS budget");
};

struct SmallVecs {
    const float* xtab; const float* bc; const float* bhh; const float* lnw; const float* lnb; float* deg; int* cls;
};

template <int H>
__device__ __forceinline__ SmallVecs stage_small(const StageX3Args& a, float* base) {
    SmallVecs v;
    float* xtab = base;
    float* bc = xtab + kMaxClsX3 * 3 * H;
    float* bhh = bc + 3 * H;
    float* lnw = bhh + 3 * H;
    float* lnb = lnw + H;
    float* deg = lnb + H;
    int* cls = reinterpret_cast<int*>(deg + kTileRows);
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    for (int i = tid; i < a.C * 3 * H; i += nt) xtab[i] = a.xtab[i];
    for (int i = tid; i < 3 * H; i += nt) { bc[i] = a.bc[i]; bhh[i] = a.bhh[i]; }
    for (int i = tid; i < H; i += nt) { lnw[i] = a.lnw ? a.lnw[i] : 1.0f; lnb[i] = a.lnb ? a.lnb[i] : 0.0f; }
    v.xtab = xtab; v.bc = bc; v.bhh = bhh; v.lnw = lnw; v.lnb = lnb; v.deg = deg; v.cls = cls;
    return v;
}


// b_hr, b_hz join the class rows of the LDS copy of xtab (the n gate keeps b_hn apart: it sits inside r * (...)).  Saves two adds
// per element in the GRU; gradients are unaffected (they are formed from the gate gradients, not from these sums).
template <int H>
__device__ __forceinline__ void fold_bhh_rz(const StageX3Args& a, const SmallVecs& sv) {
    __syncthreads();
    float* xt = const_cast<float*>(sv.xtab);
    for (int i = threadIdx.x; i < a.C * 3 * H; i += blockDim.x) { const int c = i % (3 * H); if (c < 2 * H) xt[i] += sv.bhh[c]; }
    __syncthreads();
}

// ---- neighbour-index prefetch.  The row gathers are the only HBM-latency-bound part of the kernel, so
// the CSR pointers and indices of tile t+1 are fetched while tile t is being computed and parked in LDS;
// the row phase of a tile then consists of independent loads only (own rows + up to 4 neighbour rows
// per lane group in flight at once).
// one of two LDS buffers, addressed by arithmetic on the shared-memory base (a dynamically indexed array of pointer
// structs would hide the address space from the compiler and turn every index read into a flat_load)
constexpr int kIdxStride = kPtrPad + kIdxCap + 8;
struct IdxLds { int* ptr; int* idx; __device__ int* dmax() const { return ptr + 72; } };
__device__ __forceinline__ IdxLds idx_lds(int* idx_base, int b) { return IdxLds{idx_base + b * kIdxStride, idx_base + b * kIdxStride + kPtrPad}; }

// Tile sequence of a persistent workgroup.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one:
// observed placement, used for speed only), so XCD x takes the x-th CONTIGUOUS eighth of the tiles and its workgroups
// stride through it together: the rows a tile gathers from nearby nodes (previous level of the same graph) were
// touched a few tiles earlier by the SAME XCD and are served by its L2 instead of the fabric.
struct TileSeq {
    int64_t first, end, invalid;
    int stride;
    __device__ __forceinline__ int64_t at(int k) const { const int64_t t = first + (int64_t)k * stride; return t < end ? t : invalid; }
};
__device__ __forceinline__ TileSeq tile_seq(int64_t ntiles, int xcd) {
    TileSeq s;
    s.invalid = ntiles;
    if (xcd && (gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7, per = gridDim.x >> 3;
        const int64_t chunk = (ntiles + 7) >> 3;
        s.first = x * chunk + (blockIdx.x >> 3);
        s.end = (x + 1) * chunk < ntiles ? (x + 1) * chunk : ntiles;
        s.stride = per;
        if (s.first >= s.end) s.first = s.end = ntiles;
    } else {
        s.first = blockIdx.x; s.end = ntiles; s.stride = gridDim.x;
    }
    return s;
}

__device__ __forceinline__ int ptr_prefetch(const StageX3Args& a, int64_t tile, int64_t ntiles) {
    if (tile >= ntiles || threadIdx.x > kTileRows) return 0;
    int64_t n = tile * kTileRows + threadIdx.x;
    n = n < a.N ? n : a.N;
    return a.ptr[n];
}

template <int NT>
__device__ __forceinline__ void idx_prefetch(const StageX3Args& a, const int* s_ptr, int (&ri)[kIdxCap / NT]) {
    const int e0 = s_ptr[0], ne = s_ptr[kTileRows] - e0;
#pragma unroll
    for (int k = 0; k < kIdxCap / NT; ++k) {
        const int i = threadIdx.x + k * NT;
        ri[k] = i < ne ? a.idx[e0 + i] : 0;
    }
}

template <int NT>
__device__ __forceinline__ void idx_commit(int* s_idx, const int (&ri)[kIdxCap / NT]) {
#pragma unroll
    for (int k = 0; k < kIdxCap / NT; ++k) s_idx[threadIdx.x + k * NT] = ri[k];
}

// maximum neighbour count of a tile's rows, from the CSR pointers parked in LDS (executed by wave 0)
__device__ __forceinline__ void tile_dmax(const int* s_ptr, int* s_dmax) {
    if (threadIdx.x < 64) {
        int d = s_ptr[threadIdx.x + 1] - s_ptr[threadIdx.x];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) d = max(d, __shfl_xor(d, m, 64));
        const int ne = s_ptr[kTileRows] - s_ptr[0];
        if (threadIdx.x == 0) *s_dmax = ne > kIdxCap - 8 ? (1 << 30) : d;    // index list does not fit LDS: generic path
    }
}

// Row loads of RPG tile rows per lane group: own rows first, then the neighbour lists in chunks of D
// slots per row.  Within a chunk every index is read from LDS unconditionally and every row load is
// only predicated (no waits, no dummy traffic), so RPG*D row loads (x2 with DY) are in flight per lane.
// The chunk loop's trip count is the tile's maximum degree: workgroup-uniform.
template <int H, int D, int RPG, bool DY>
__device__ __forceinline__ void rows_chunked(const StageX3Args& a, int64_t base, int grp, int groups, int lr, const int* s_ptr,
                                             const int* s_idx, int dmax, bool two, float4 (&acc)[RPG], float4 (&own)[RPG],
                                             float4 (&dy)[RPG], float (&deg)[RPG], int (&cls)[RPG]) {
    // caller guarantees: the whole tile lies inside [0, N) and its index list inside LDS
    const int e0t = s_ptr[0];
    int rel0[RPG], d[RPG];
#pragma unroll
    for (int rr = 0; rr < RPG; ++rr) {
        const int row = grp + rr * groups;
        const int p0 = s_ptr[row];
        rel0[rr] = p0 - e0t;
        d[rr] = s_ptr[row + 1] - p0;
    }
    // chunk 0 is peeled so that its row loads are issued together with the own-row loads
    int j0[RPG][D];
#pragma unroll
    for (int rr = 0; rr < RPG; ++rr)
#pragma unroll
        for (int k = 0; k < D; ++k) j0[rr][k] = s_idx[min(rel0[rr] + k, kIdxCap + 7)];
    f32x4 v0[RPG][D], g0[RPG][D];
#pragma unroll
    for (int rr = 0; rr < RPG; ++rr) {
        const int64_t node = base + grp + rr * groups;
        own[rr] = ld4(a.h_in + (a.own_idx ? (int64_t)a.own_idx[node] : node) * H + 4 * lr);
        if (DY) dy[rr] = ld4(a.gy_direct + node * H + 4 * lr); else dy[rr] = zero4();
        cls[rr] = a.xcls[node];
        // predicated loads into registers that are NOT written on the untaken path (a zero-initialised
        // destination would be a write-after-write hazard and make the compiler drain vmcnt per load)
#pragma unroll
        for (int k = 0; k < D; ++k)
            if (k < d[rr]) {
                v0[rr][k] = *reinterpret_cast<const f32x4*>(a.h_in + (int64_t)((unsigned)j0[rr][k] >> a.hshift) * H + 4 * lr);
                if (DY && two) g0[rr][k] = *reinterpret_cast<const f32x4*>(a.gy_agg + (int64_t)(j0[rr][k] & a.gmask) * H + 4 * lr);
            }
    }
#pragma unroll
    for (int rr = 0; rr < RPG; ++rr) {
        deg[rr] = (float)d[rr];
        acc[rr] = zero4();
#pragma unroll
        for (int k = 0; k < D; ++k)
            if (k < d[rr]) {
                acc[rr] = add4(acc[rr], make_float4(v0[rr][k][0], v0[rr][k][1], v0[rr][k][2], v0[rr][k][3]));
                if (DY && two) dy[rr] = add4(dy[rr], make_float4(g0[rr][k][0], g0[rr][k][1], g0[rr][k][2], g0[rr][k][3]));
            }
    }
    for (int c0 = D; c0 < dmax; c0 += D) {
        int j[RPG][D];
#pragma unroll
        for (int rr = 0; rr < RPG; ++rr)
#pragma unroll
            for (int k = 0; k < D; ++k) j[rr][k] = s_idx[min(rel0[rr] + c0 + k, kIdxCap + 7)];
        f32x4 v[RPG][D], g[RPG][D];
#pragma unroll
        for (int rr = 0; rr < RPG; ++rr)
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (c0 + k < d[rr]) {
                    v[rr][k] = *reinterpret_cast<const f32x4*>(a.h_in + (int64_t)((unsigned)j[rr][k] >> a.hshift) * H + 4 * lr);
                    if (DY && two) g[rr][k] = *reinterpret_cast<const f32x4*>(a.gy_agg + (int64_t)(j[rr][k] & a.gmask) * H + 4 * lr);
                }
#pragma unroll
        for (int rr = 0; rr < RPG; ++rr)
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (c0 + k < d[rr]) {
                    acc[rr] = add4(acc[rr], make_float4(v[rr][k][0], v[rr][k][1], v[rr][k][2], v[rr][k][3]));
                    if (DY && two) dy[rr] = add4(dy[rr], make_float4(g[rr][k][0], g[rr][k][1], g[rr][k][2], g[rr][k][3]));
                }
    }
}

// generic degree: per-row loops (rare tiles with a high fan-out node or an index list beyond LDS).  A row with more than kHeavyRow
// neighbours takes its sums from the pre-pass (k_heavy_sums) when the caller listed it: walked here, one neighbour after the other
// by a single lane group, a 100,000-consumer net costs 38 ms per launch.
constexpr int kHeavyRow = 64;

template <int H, bool DY>
__device__ __forceinline__ void row_generic(const StageX3Args& a, int64_t node, int row, int lr, const int* s_ptr, bool two,
                                            float4& acc, float4& own, float4& dy, float& deg, int& cls) {
    acc = zero4(); own = zero4(); dy = zero4(); deg = 0.f; cls = 0;
    if (node >= a.N) return;
    const int p0 = s_ptr[row], p1 = s_ptr[row + 1];
    deg = (float)(p1 - p0);
    own = ld4(a.h_in + (a.own_idx ? (int64_t)a.own_idx[node] : node) * H + 4 * lr);
    if (DY) dy = ld4(a.gy_direct + node * H + 4 * lr);
    cls = a.xcls[node];
    if (p1 - p0 > kHeavyRow && a.heavy_n > 0) {
        int lo = 0, hi = a.heavy_n - 1, k = -1;
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1, v = a.heavy_nodes[mid];
            if (v == (int)node) { k = mid; break; }
            if (v < (int)node) lo = mid + 1; else hi = mid - 1;
        }
        if (k >= 0) {
            acc = ld4(a.heavy_a + (int64_t)k * H + 4 * lr);
            if (DY && two) dy = add4(dy, ld4(a.heavy_g + (int64_t)k * H + 4 * lr));
            return;
        }
    }
    for (int e = p0; e < p1; ++e) {
        const int jv = a.idx[e];
        acc = add4(acc, ld4(a.h_in + (int64_t)((unsigned)jv >> a.hshift) * H + 4 * lr));
        if (DY && two) dy = add4(dy, ld4(a.gy_agg + (int64_t)(jv & a.gmask) * H + 4 * lr));
    }
}

// Neighbour sums of the listed heavy rows, one 256-thread workgroup per row: lane group g takes neighbours g, g + 16, ... (four in
// flight), the 16 partial sums meet in LDS and are added in group order (deterministic).  out_a[k] = sum h_in[nbr], out_g[k] = sum
// gy_agg[nbr] (gy_agg nullable).
template <int H>
static __global__ __launch_bounds__(256) void k_heavy_sums(int K, const int32_t* nodes, const int32_t* ptr, const int32_t* idx,
                                                           const float* src_a, const float* src_g, float* out_a, float* out_g, int hshift, int gmask) {
    constexpr int LPR = H / 4, G = 256 / LPR;
    __shared__ __attribute__((aligned(16))) float s_p[2][G][H];
    const int lr = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    for (int k = blockIdx.x; k < K; k += gridDim.x) {
        const int node = nodes[k];
        const int p0 = ptr[node], p1 = ptr[node + 1];
        float4 sa = zero4(), sg = zero4();
        int e = p0 + grp;
        for (; e + 3 * G < p1; e += 4 * G) {
            int j[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) j[u] = idx[e + u * G];
            float4 va[4], vg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                va[u] = ld4(src_a + (int64_t)((unsigned)j[u] >> hshift) * H + 4 * lr);
                if (src_g) vg[u] = ld4(src_g + (int64_t)(j[u] & gmask) * H + 4 * lr);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sa = add4(sa, va[u]);
                if (src_g) sg = add4(sg, vg[u]);
            }
        }
        for (; e < p1; e += G) {
            const int jv = idx[e];
            sa = add4(sa, ld4(src_a + (int64_t)((unsigned)jv >> hshift) * H + 4 * lr));
            if (src_g) sg = add4(sg, ld4(src_g + (int64_t)(jv & gmask) * H + 4 * lr));
        }
        __syncthreads();
        st4(&s_p[0][grp][4 * lr], sa);
        st4(&s_p[1][grp][4 * lr], sg);
        __syncthreads();
        if (grp == 0) {
            float4 ta = zero4(), tg = zero4();
            for (int g = 0; g < G; ++g) { ta = add4(ta, ld4(&s_p[0][g][4 * lr])); tg = add4(tg, ld4(&s_p[1][g][4 * lr])); }
            st4(out_a + (int64_t)k * H + 4 * lr, ta);
            if (src_g) st4(out_g + (int64_t)k * H + 4 * lr, tg);
        }
    }
}

// pre-pass of a stage launch: fills a.heavy_a / a.heavy_g from `ws` ([2][heavy_n][H] floats); no-op without heavy rows
template <int H>
static inline void launch_heavy_sums(StageX3Args& a, int heavy_n, const int32_t* heavy_nodes, float* ws, bool with_g, hipStream_t st) {
    a.heavy_n = 0;
    if (heavy_n <= 0 || heavy_nodes == nullptr || ws == nullptr) return;
    float* out_a = ws;
    float* out_g = ws + (int64_t)heavy_n * H;
    const float* src_g = with_g ? a.gy_agg : nullptr;
    hipLaunchKernelGGL(k_heavy_sums<H>, dim3(heavy_n < 1024 ? heavy_n : 1024), dim3(256), 0, st, heavy_n, heavy_nodes, a.ptr, a.idx, a.h_in, src_g, out_a, out_g, a.hshift, a.gmask);
    a.heavy_n = heavy_n; a.heavy_nodes = heavy_nodes; a.heavy_a = out_a; a.heavy_g = out_g;
}

template <int H, int RPG, bool DY, int D = 2>
__device__ __forceinline__ void tile_rows(const StageX3Args& a, int64_t base, int grp, int groups, int lr, const int* s_ptr,
                                          const int* s_idx, int dmax, float4 (&acc)[RPG], float4 (&own)[RPG], float4 (&dy)[RPG],
                                          float (&deg)[RPG], int (&cls)[RPG]) {
    const bool two = DY && a.gy_agg != nullptr;
    // D neighbour slots per row and chunk: RPG*D (x2 with DY) row loads in flight per lane
    if (dmax < (1 << 30) && base + kTileRows <= a.N) {
        rows_chunked<H, D, RPG, DY>(a, base, grp, groups, lr, s_ptr, s_idx, dmax, two, acc, own, dy, deg, cls);
    } else {                                         // partial last tile, or an index list beyond LDS
#pragma unroll
        for (int rr = 0; rr < RPG; ++rr) {
            const int row = grp + rr * groups;
            row_generic<H, DY>(a, base + row, row, lr, s_ptr, two, acc[rr], own[rr], dy[rr], deg[rr], cls[rr]);
        }
    }
}


// H = 64 with 8 waves: the LDS read rate bounds the weight gradient, so every wave owns a 2x2 block of output tiles
// of ONE matrix (waves 0-3: Wc against the aggregate planes, waves 4-7: Whh against the own-row planes) and reuses
// each transposed fragment twice: 8 fragment reads per 12 MFMA triples instead of 12.
template <int H>
__device__ __forceinline__ void wgrad_blk_x3(f32x4 (&acc)[4], const __bf16* d_hi, const __bf16* d_lo, const __bf16* x_hi, const __bf16* x_lo) {
    constexpr int LDP = H + 8;
    const int b = (threadIdx.x >> 6) & 3;
    const int it0 = 2 * (b >> 1), jt0 = 2 * (b & 1);
#pragma unroll 1
    for (int kk = 0; kk < kTileRows / 32; ++kk) {
        const int k0 = 32 * kk;
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { bh[j] = ldfrag_tr2(x_hi, LDP, k0, (jt0 + j) * 16); bl[j] = ldfrag_tr2(x_lo, LDP, k0, (jt0 + j) * 16); }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16x8 ah = ldfrag_tr2(d_hi, LDP, k0, (it0 + i) * 16), al = ldfrag_tr2(d_lo, LDP, k0, (it0 + i) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) mma_x3(acc[i * 2 + j], ah, al, bh[j], bl[j]);
        }
    }
}

template <int H>
__device__ __forceinline__ void wgrad_blk_flush_x3(const f32x4 (&acc)[4], float* dW) {
    const int lane = threadIdx.x & 63, b = (threadIdx.x >> 6) & 3, r = lane & 15, q = lane >> 4;
    const int it0 = 2 * (b >> 1), jt0 = 2 * (b & 1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dW + (int64_t)((it0 + i) * 16 + q * 4 + e) * H + (jt0 + j) * 16 + r, acc[i * 2 + j][e]);
}


}  // namespace mgv
