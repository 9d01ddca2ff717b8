// Backward of a structural-encoder half round (digae_layer.py:266-275 under autograd), third decomposition, H = 64:
// the two waves of every SIMD run DIFFERENT programs on different tiles, so that one wave's vector work runs beside the
// other's matrix / memory work (the second decomposition ran all eight waves through the same phase at the same time:
// its MFMA, VALU, row-gather, LDS-fragment and store times simply added up — measured by ablation, profiles/r03_*).
//
//   * 32-row tiles, a two-stage pipeline inside the 8-wave workgroup.  During period p
//       the FRONT waves 0-3 (one per SIMD) work on tile p:    recompute of BOTH matrices' pre-activations for their 16 hidden
//           columns (wave wc owns column tile wc of Wc and of Whh: the m0/m1 exchange of the second decomposition is gone),
//           GRU forward, LayerNorm partials | barrier A | LayerNorm + GRU backward, gate-gradient planes, dh*z;
//       the BACK waves 4-7 (their SIMD partners) work on tile p-1 and p+1:  row gather of tile p+1 (they are the only waves
//           that wait for memory), weight gradients of tile p-1 | barrier A | operand planes of tile p+1, dgrad of tile p-1
//           with the outputs stored straight from the accumulators.
//     Two workgroup barriers per 32 rows (A: LayerNorm partials complete, tile p-1's operand planes free; B: gate-gradient
//     planes of tile p and operand planes of tile p+1 complete).
//   * LDS: operand planes (agg/hin, hi/lo) x2 buffers, gate-gradient planes (r, z, n, n*r; hi/lo) x2, dh*z x2, [deg, onehot, 1]
//     planes x2, one dY tile, LayerNorm partials, neighbour-index ring: 152 KB.
//   * Front waves keep their 24 recompute fragments (96 VGPRs) for the whole kernel and never touch global memory except for
//     the L2 prefetch of tile p+2's rows.  Back waves keep the weight-gradient accumulators of both matrices (112 VGPRs) and
//     stream their dgrad fragments from L2 (24 KB per wave and tile, 24 VGPRs at a time, issued a phase ahead of use).
//   * every product is transposed as in the second decomposition (weights as the MFMA A operand): a lane holds ONE node and four
//     consecutive hidden columns.  Parameter gradients leave through the same per-workgroup slabs and fixed-order reduction:
//     no float atomics, bit-identical from run to run.
#include "struct_stage_x3_common.h"

namespace mgv {

struct B3 {
    static constexpr int H = 64;
    static constexpr int TR = 32;                             // rows per tile
    static constexpr int LDP = H + 8;                         // bf16 elements per plane row
    static constexpr int PE = TR * LDP;                       // elements of one plane
    static constexpr int PB = PE * 2;                         // bytes of one plane
    static constexpr int LDF = H + 4;                         // floats per fp32 tile row
    static constexpr int IDXCAP = 256;                        // neighbour entries of one tile kept in LDS
    static constexpr int PTRPAD = 48;                         // 33 CSR pointers + the tile's maximum degree at [40]
    static constexpr int IDXSTRIDE = PTRPAD + IDXCAP + 8;
    static constexpr int SMALL_F = kMaxClsX3 * 3 * H + 3 * H + 3 * H + H + H + kTileRows + kTileRows;     // stage_small layout
    static constexpr int o_x = 0;                             // 2 buffers x {agg_hi, agg_lo, hin_hi, hin_lo}
    static constexpr int o_dg = o_x + 2 * 4 * PB;             // 2 buffers x 4 planes x {hi, lo}
    static constexpr int o_dy = o_dg + 2 * 8 * PB;            // fp32 dY tile (read by P2 only: the front waves keep it in registers)
    static constexpr int o_dhz = o_dy + TR * LDF * 4;         // 2 buffers x [4 waves][2 row tiles][64 lanes] float4
    static constexpr int o_small = o_dhz + 2 * 4 * 2 * 1024;
    static constexpr int o_part = o_small + SMALL_F * 4;      // LayerNorm partials [32 rows][4 column waves] float4
    static constexpr int o_idx = o_part + TR * 4 * 16;
    static constexpr int o_xe = o_idx + 3 * IDXSTRIDE * 4;    // ring of three tiles: being gathered, being prefetched into L2, being filled    // 2 buffers x {xe_hi, xe_lo} [32][XLD]
    static constexpr int o_lnacc = o_xe + 2 * 2 * TR * XLD * 2;       // per front wave LayerNorm affine gradient sums [4][2][16]
    static constexpr int bytes = o_lnacc + 4 * 2 * 16 * 4;
    static_assert(bytes <= 160 * 1024, "LDS budget");
    // per-workgroup gradient slab: the second decomposition's layout ([8 = m*4 + wc][14 float4 slots][64 lanes], dlnw[64], dlnb[64]);
    // back wave wc fills both of its matrices' blocks and k_struct_stage_bwd2_reduce sums it unchanged
    static constexpr int SLOTS = 14;
    static constexpr int SLAB_W = 8 * SLOTS * 64 * 4;
    static constexpr int SLAB = SLAB_W + 2 * H;
};

struct B3Args {
    StageX3Args s;
    float* slab;        // [gridDim][B3::SLAB]
};

int launch_stage_slab_reduce(const StageX3Args& s, float* workspace, int grid, hipStream_t st);    // struct_stage_bwd2_x3.hip

struct IdxL3 { int* ptr; int* idx; __device__ int* dmax() const { return ptr + 40; } };
__device__ __forceinline__ IdxL3 idxl3(int* base, int b) { return IdxL3{base + b * B3::IDXSTRIDE, base + b * B3::IDXSTRIDE + B3::PTRPAD}; }

__device__ __forceinline__ float quad_rows_sum3(float v) {     // sum over the four lanes r, r+16, r+32, r+48
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// CSR pointer of row bt of a 32-row tile (33 values; bt = thread index inside the back half)
__device__ __forceinline__ int ptr_prefetch3(const StageX3Args& a, int64_t tile, int64_t ntiles, int bt) {
    if (tile >= ntiles || bt > B3::TR) return 0;
    int64_t n = tile * B3::TR + bt;
    n = n < a.N ? n : a.N;
    return a.ptr[n];
}

// Row loads of a 32-row tile by the 16 lane groups of the back half, two rows per group, in two steps so that the loads fly
// while the weight-gradient MFMAs run: `rows3_issue` requests the own rows, the dY rows and the first D neighbour slots of both
// rows (every row load only predicated: no waits, no dummy traffic); `rows3_finish` sums them and walks the remaining chunks
// (the trip count is the tile's maximum degree: workgroup-uniform).  rows_chunked of struct_stage_x3_common.h for this tile size.
template <int D>
struct Rows3 {
    f32x4 own[2], dy[2], v0[2][D], g0[2][D];
    int rel0[2], d[2], cls[2];
};

template <int D>
__device__ __forceinline__ void rows3_issue(const StageX3Args& a, int64_t base, int grp, int lr, const int* s_ptr, const int* s_idx, bool two, Rows3<D>& R) {
    constexpr int H = B3::H, CAP = B3::IDXCAP;
    const int e0t = s_ptr[0];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = grp + rr * 16;
        const int p0 = s_ptr[row];
        R.rel0[rr] = p0 - e0t;
        R.d[rr] = s_ptr[row + 1] - p0;
    }
    int j0[2][D];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int k = 0; k < D; ++k) j0[rr][k] = s_idx[min(R.rel0[rr] + k, CAP + 7)];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int64_t node = base + grp + rr * 16;
        R.own[rr] = *reinterpret_cast<const f32x4*>(a.h_in + (a.own_idx ? (int64_t)a.own_idx[node] : node) * H + 4 * lr);
        R.dy[rr] = *reinterpret_cast<const f32x4*>(a.gy_direct + node * H + 4 * lr);
        R.cls[rr] = a.xcls[node];
#pragma unroll
        for (int k = 0; k < D; ++k)
            if (k < R.d[rr]) {
                R.v0[rr][k] = *reinterpret_cast<const f32x4*>(a.h_in + (int64_t)((unsigned)j0[rr][k] >> a.hshift) * H + 4 * lr);
                if (two) R.g0[rr][k] = *reinterpret_cast<const f32x4*>(a.gy_agg + (int64_t)(j0[rr][k] & a.gmask) * H + 4 * lr);
            }
    }
}

template <int D>
__device__ __forceinline__ void rows3_finish(const StageX3Args& a, int lr, const int* s_idx, int dmax, bool two, const Rows3<D>& R,
                                             float4 (&acc)[2], float4 (&own)[2], float4 (&dy)[2], float (&deg)[2], int (&cls)[2]) {
    constexpr int H = B3::H, CAP = B3::IDXCAP;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        deg[rr] = (float)R.d[rr];
        cls[rr] = R.cls[rr];
        own[rr] = make_float4(R.own[rr][0], R.own[rr][1], R.own[rr][2], R.own[rr][3]);
        dy[rr] = make_float4(R.dy[rr][0], R.dy[rr][1], R.dy[rr][2], R.dy[rr][3]);
        acc[rr] = zero4();
#pragma unroll
        for (int k = 0; k < D; ++k)
            if (k < R.d[rr]) {
                acc[rr] = add4(acc[rr], make_float4(R.v0[rr][k][0], R.v0[rr][k][1], R.v0[rr][k][2], R.v0[rr][k][3]));
                if (two) dy[rr] = add4(dy[rr], make_float4(R.g0[rr][k][0], R.g0[rr][k][1], R.g0[rr][k][2], R.g0[rr][k][3]));
            }
    }
    for (int c0 = D; c0 < dmax; c0 += D) {
        int j[2][D];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int k = 0; k < D; ++k) j[rr][k] = s_idx[min(R.rel0[rr] + c0 + k, CAP + 7)];
        f32x4 v[2][D], g[2][D];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (c0 + k < R.d[rr]) {
                    v[rr][k] = *reinterpret_cast<const f32x4*>(a.h_in + (int64_t)((unsigned)j[rr][k] >> a.hshift) * H + 4 * lr);
                    if (two) g[rr][k] = *reinterpret_cast<const f32x4*>(a.gy_agg + (int64_t)(j[rr][k] & a.gmask) * H + 4 * lr);
                }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (c0 + k < R.d[rr]) {
                    acc[rr] = add4(acc[rr], make_float4(v[rr][k][0], v[rr][k][1], v[rr][k][2], v[rr][k][3]));
                    if (two) dy[rr] = add4(dy[rr], make_float4(g[rr][k][0], g[rr][k][1], g[rr][k][2], g[rr][k][3]));
                }
    }
}

// whole gather of a tile in one go (prologue), and the generic path (partial last tile, or an index list beyond LDS: heavy rows
// take their pre-pass sums)
__device__ __forceinline__ bool rows3_chunked(const StageX3Args& a, int64_t base, int dmax) { return dmax < (1 << 30) && base + B3::TR <= a.N; }

__device__ __forceinline__ void tile_rows3(const StageX3Args& a, int64_t base, int grp, int lr, const int* s_ptr, const int* s_idx, int dmax,
                                           float4 (&acc)[2], float4 (&own)[2], float4 (&dy)[2], float (&deg)[2], int (&cls)[2]) {
    const bool two = a.gy_agg != nullptr;
    if (rows3_chunked(a, base, dmax)) {
        Rows3<2> R;
        rows3_issue<2>(a, base, grp, lr, s_ptr, s_idx, two, R);
        rows3_finish<2>(a, lr, s_idx, dmax, two, R, acc, own, dy, deg, cls);
    } else {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = grp + rr * 16;
            row_generic<B3::H, true>(a, base + row, row, lr, s_ptr, two, acc[rr], own[rr], dy[rr], deg[rr], cls[rr]);
        }
    }
}

// dgrad of ONE matrix for the two row tiles of a 32-row tile: out[i] (+= seed) = sum over the 192 gate columns of W^T dG.
// The six k-step fragment pairs of the wave's output column tile come from L2 (wd = hi block; lo block at +BLK): `dgrad_load3` requests
// them a phase ahead of `dgrad_mul3` (one wave per SIMD runs this phase: nobody else hides its L2 latency).
struct DgW3 { bf16x8 hi[6], lo[6]; };
template <int K0, int K1>
__device__ __forceinline__ void dgrad_load3(DgW3& w, const __bf16* wd, int lane) {
    constexpr int BLK = 3 * B3::H * B3::H;
#pragma unroll
    for (int k = K0; k < K1; ++k) { w.hi[k] = ldfrag_global(wd + k * 512 + lane * 8); w.lo[k] = ldfrag_global(wd + BLK + k * 512 + lane * 8); }
}
// 12 steps (k-step, row tile) of 3 MFMAs; step s + 1's two operand fragments are read from LDS while step s multiplies, and the
// scheduler may not pull later reads forward (register budget).  Steps 0-5: gate columns 0..95 (k-steps 0-2: planes r, r, z), both
// row tiles; steps 6-11: gate columns 96..191 (k-steps 3-5: planes z, n | n*r) row tile by row tile, each finished accumulator
// stored at once, straight from the registers: lane (r, q) holds 16 contiguous bytes of row 16 i + r.
// pn = the n plane (Wc) or the n*r plane (Whh) of the gate-gradient buffer dgp.
__device__ __forceinline__ void dgrad_mul3(const DgW3& w, const __bf16* dgp, const __bf16* pn, f32x4 (&dgo)[2], float* go, int64_t base, int64_t N,
                                           int r, int q, int c0) {
    constexpr int H = B3::H, LDP = B3::LDP, PE = B3::PE;
    auto frag_ptr = [&](int s_) -> const __bf16* {
        const int k = s_ < 6 ? s_ >> 1 : 3 + (s_ - 6) % 3, i = s_ < 6 ? s_ & 1 : (s_ - 6) / 3;
        const __bf16* ph = k < 2 ? dgp : (k < 4 ? dgp + 2 * PE : pn);
        return ph + (i * 16 + r) * LDP + 32 * (k & 1) + 8 * q;
    };
    bf16x8 fh[2], fl[2];
    fh[0] = ldfrag(frag_ptr(0)); fl[0] = ldfrag(frag_ptr(0) + PE);
#pragma unroll
    for (int s_ = 0; s_ < 12; ++s_) {
        if (s_ + 1 < 12) { fh[(s_ + 1) & 1] = ldfrag(frag_ptr(s_ + 1)); fl[(s_ + 1) & 1] = ldfrag(frag_ptr(s_ + 1) + PE); }
        const int k = s_ < 6 ? s_ >> 1 : 3 + (s_ - 6) % 3, i = s_ < 6 ? s_ & 1 : (s_ - 6) / 3;
        mma_x3(dgo[i], w.hi[k], w.lo[k], fh[s_ & 1], fl[s_ & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (s_ == 8 || s_ == 11) {
            const int64_t node = base + i * 16 + r;
            if (node < N) *reinterpret_cast<f32x4*>(go + node * H + c0) = dgo[i];
        }
    }
}

__global__ __launch_bounds__(kThreadsX3) void k_struct_stage_bwd3_x3(B3Args args) {
    const StageX3Args& a = args.s;
    constexpr int H = B3::H, TR = B3::TR, LDP = B3::LDP, LDF = B3::LDF, PE = B3::PE, BLK = 3 * H * H;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* s_x = reinterpret_cast<__bf16*>(smem_raw + B3::o_x);            // buffer b at b * 4 PE: agg_hi, agg_lo, hin_hi, hin_lo
    __bf16* s_dg = reinterpret_cast<__bf16*>(smem_raw + B3::o_dg);          // buffer b at b * 8 PE: plane p (r, z, n, n*r) hi at 2p PE, lo at (2p + 1) PE
    float* s_dy = reinterpret_cast<float*>(smem_raw + B3::o_dy);
    f32x4* s_dhz = reinterpret_cast<f32x4*>(smem_raw + B3::o_dhz);          // buffer b at b * 512: [wc][row tile][lane]
    const SmallVecs sv = stage_small<H>(a, reinterpret_cast<float*>(smem_raw + B3::o_small));
    fold_bhh_rz<H>(a, sv);
    f32x4* s_part = reinterpret_cast<f32x4*>(smem_raw + B3::o_part);
    int* idx_base = reinterpret_cast<int*>(smem_raw + B3::o_idx);
    __bf16* s_xe = reinterpret_cast<__bf16*>(smem_raw + B3::o_xe);          // buffer b at b * 2 * TR * XLD: xe_hi, xe_lo
    float* s_lnacc = reinterpret_cast<float*>(smem_raw + B3::o_lnacc);

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wc = w & 3;
    const bool back = w >= 4;
    // Lane-derived indices are re-derived per phase behind an opaque copy of the thread id: otherwise every LDS address of the loop
    // body is hoisted out of the tile loop and the loop-invariant address registers are spilled (struct_stage_bwd2_x3.hip).
#define LANE_IDS3 \
    int t_ = tid; asm volatile("" : "+v"(t_)); \
    const int lane = t_ & 63, r = lane & 15, q = lane >> 4, c0 = 16 * wc + 4 * q; \
    (void)lane; (void)r; (void)q; (void)c0;
    const bool has_ln = a.lnw != nullptr;
    const bool need_dgrad = a.g_direct_out != nullptr;
    const int64_t ntiles = (a.N + TR - 1) / TR;
    const TileSeq seq = tile_seq(ntiles, a.xcd);
    const int lane0 = tid & 63;

    for (int i = tid; i < 4 * 2 * 16; i += kThreadsX3) s_lnacc[i] = 0.f;

    // ---- prologue (back waves): pointers / indices of this workgroup's first two tiles, operand planes of the first
    int rp = 0;                                   // back: CSR pointer of a tile two periods ahead
    int ri = 0;                                   // back: neighbour index being moved to LDS
    const int bt = tid - 256;                     // thread index inside the back half
    if (back) {
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
            rp = ptr_prefetch3(a, seq.at(bb), ntiles, bt);
            if (bt <= TR) idxl3(idx_base, bb).ptr[bt] = rp;
        }
    }
    __syncthreads();
    if (back) {
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
            const int* sp = idxl3(idx_base, bb).ptr;
            const int e0 = sp[0], ne = sp[TR] - e0;
            idxl3(idx_base, bb).idx[bt] = bt < ne ? a.idx[e0 + bt] : 0;
            if (bt < 8) idxl3(idx_base, bb).idx[B3::IDXCAP + bt] = 0;
            if (w == 4) {
                int d = lane0 < TR ? sp[lane0 + 1] - sp[lane0] : 0;
#pragma unroll
                for (int mm = 32; mm >= 1; mm >>= 1) d = max(d, __shfl_xor(d, mm, 64));
                if (lane0 == 0) *idxl3(idx_base, bb).dmax() = ne > B3::IDXCAP - 8 ? (1 << 30) : d;
            }
        }
        rp = ptr_prefetch3(a, seq.at(3), ntiles, bt);
    }
    __syncthreads();

    // operand planes of one gathered tile -> x buffer xb, dY tile, deg/cls, xe buffer xb   (back waves).  Two steps: `prep` turns the
    // gathered values into what LDS will hold (the split into bf16 hi/lo: every wait for the gather lands here, in interval 1) and
    // `store` only issues LDS stores (interval 2: no memory wait can land behind the dgrad weight loads issued there).
    struct RowPack { bf16x4 ah[2], al[2], oh[2], ol[2]; float4 dy[2]; float deg[2]; int cls[2]; __bf16 xh[2], xl[2]; };
    auto prep_rows = [&](RowPack& rp_, const float4 (&acc)[2], const float4 (&own)[2], const float4 (&dy)[2], const float (&deg)[2], const int (&cls)[2]) {
        int t_ = tid; asm volatile("" : "+v"(t_));
        const int lr = t_ & 15;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            split4(acc[rr], rp_.ah[rr], rp_.al[rr]);
            split4(own[rr], rp_.oh[rr], rp_.ol[rr]);
            rp_.dy[rr] = dy[rr]; rp_.deg[rr] = deg[rr]; rp_.cls[rr] = cls[rr];
            const float xe = lr == 0 ? deg[rr] : (lr <= 8 ? (cls[rr] == lr - 1 ? 1.0f : 0.0f) : (lr == 9 ? 1.0f : 0.0f));
            split_bf16(xe, rp_.xh[rr], rp_.xl[rr]);
        }
    };
    auto store_rows = [&](int xb, const RowPack& rp_) {
        int t_ = tid; asm volatile("" : "+v"(t_));
        const int grp = (t_ - 256) >> 4, lr = t_ & 15;
        __bf16* xp = s_x + xb * 4 * PE;
        __bf16* xe_hi = s_xe + xb * 2 * TR * XLD;
        __bf16* xe_lo = xe_hi + TR * XLD;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = grp + rr * 16;
            st_bf4(xp + row * LDP + 4 * lr, rp_.ah[rr]); st_bf4(xp + PE + row * LDP + 4 * lr, rp_.al[rr]);
            st_bf4(xp + 2 * PE + row * LDP + 4 * lr, rp_.oh[rr]); st_bf4(xp + 3 * PE + row * LDP + 4 * lr, rp_.ol[rr]);
            st4(s_dy + row * LDF + 4 * lr, rp_.dy[rr]);
            if (lr == 0) { sv.deg[row] = rp_.deg[rr]; sv.cls[row] = rp_.cls[rr]; }
            xe_hi[row * XLD + lr] = rp_.xh[rr]; xe_lo[row * XLD + lr] = rp_.xl[rr];
        }
    };

    if (back) {
        if (seq.at(0) < ntiles) {
            int t_ = tid; asm volatile("" : "+v"(t_));
            const int grp = (t_ - 256) >> 4, lr = t_ & 15;
            float4 acc[2], own[2], dy[2];
            float deg[2];
            int cls[2];
            tile_rows3(a, seq.at(0) * TR, grp, lr, idxl3(idx_base, 0).ptr, idxl3(idx_base, 0).idx, *idxl3(idx_base, 0).dmax(), acc, own, dy, deg, cls);
            RowPack pk;
            prep_rows(pk, acc, own, dy, deg, cls);
            store_rows(0, pk);
        }
    }
    lds_barrier();

    if (!back) {
        // =================================================================================== FRONT waves 0-3
        // recompute fragments of both matrices for column tile wc: [m][k-step][gate], hi and lo: 96 VGPRs, resident
        bf16x8 wr_hi[2][2][3], wr_lo[2][2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const __bf16* ph = a.wpack + (2 * m) * BLK + wc * 2 * 512 + lane0 * 8;
            const __bf16* pl = a.wpack + (2 * m + 1) * BLK + wc * 2 * 512 + lane0 * 8;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    wr_hi[m][ks][g] = ldfrag_global(ph + (g * 8 + ks) * 512);
                    wr_lo[m][ks][g] = ldfrag_global(pl + (g * 8 + ks) * 512);
                }
        }
        int s2 = 2;                               // ring slot of tile p + 2
        STAMP_DECL
        for (int p = 0;; ++p, s2 = s2 == 2 ? 0 : s2 + 1) {
            const int64_t tF = seq.at(p), tB = p >= 1 ? seq.at(p - 1) : ntiles;
            if (tF >= ntiles && tB >= ntiles) break;
            STAMP_BEGIN;
            const bool act = tF < ntiles;
            const int xb = p & 1;
            const __bf16* xp = s_x + xb * 4 * PE;
            // ---- L2 prefetch of tile p+2's rows (its pointers / indices were parked in LDS during period p-1 ... p): one 4-byte load
            //      per 128-byte line; the loaded words only feed a comparison that never holds.  The back waves gather that tile
            //      one period later (during period p+1) from the XCD's L2.  Front waves issue no other global load, so nothing queues behind these.
            unsigned pf0 = 0, pf1 = 0, pf2 = 0;
            if (a.prefetch && seq.at(p + 2) < ntiles) {
                // the ring slot of tile p+2 is complete since barrier B of period p-1 (the prologue for p = 0)
                const int64_t nb = seq.at(p + 2) * TR;
                const int* n_ptr = idxl3(idx_base, s2).ptr;
                const int* n_idx = idxl3(idx_base, s2).idx;
                const int ne2 = 2 * min(n_ptr[TR] - n_ptr[0], B3::IDXCAP);
                const bool two = a.gy_agg != nullptr;
                const unsigned nbytes = (unsigned)min((int64_t)a.N * H * 4, (int64_t)0xfffffff0);
                const auto rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.h_in), 0, nbytes, 0x00020000);
                const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy_direct), 0, nbytes, 0x00020000);
                if (w < 2) {                                     // own rows (waves 0) and dY rows (wave 1): 32 rows x 2 lines each
                    int64_t row = nb + (lane0 >> 1);
                    row = row < a.N ? row : a.N - 1;
                    if (w == 1 || a.hshift == 0) pf0 = __builtin_amdgcn_raw_buffer_load_b32(w == 0 ? rh : rd, (unsigned)(row * (H * 4)) + 128u * (lane0 & 1), 0, 0);
                } else {                                         // neighbour rows of h_in: entries 0..63 (waves 2-3)
                    const int t = tid - 128;
                    if (t < ne2 && a.hshift == 0) pf0 = __builtin_amdgcn_raw_buffer_load_b32(rh, (unsigned)n_idx[t >> 1] * (H * 4) + 128u * (t & 1), 0, 0);
                }
                if (two && tid < ne2) {                          // neighbour rows of gy_agg: entries 0..127
                    const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy_agg), 0, nbytes, 0x00020000);
                    pf1 = __builtin_amdgcn_raw_buffer_load_b32(rg, (unsigned)(n_idx[tid >> 1] & a.gmask) * (H * 4) + 128u * (tid & 1), 0, 0);
                }
                if (tid + 128 < ne2 && a.hshift == 0)            // neighbour rows of h_in: entries 64..191
                    pf2 = __builtin_amdgcn_raw_buffer_load_b32(rh, (unsigned)n_idx[(tid + 128) >> 1] * (H * 4) + 128u * (tid & 1), 0, 0);
            }
            // ---- interval 1: P1 (pre-activations of both matrices, transposed: lane (r, q) <- node 16 il + r, columns c0..c0+3)
            //      and P2 (GRU forward values, LayerNorm partials over this wave's 16 columns)
            f32x4 vr[2], vz[2], vn[2], vg[2], vd[2], vh[2], vdy[2];      // r, z, n, Whh_n h + b_hn, pre - (wave mean), own row, dY
            float mw[2];
            if (act) {
                f32x4 oa[2][3][2];
                {
                    LANE_IDS3
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const __bf16* xh = xp + (2 * m) * PE;
                        const __bf16* xl = xh + PE;
#pragma unroll
                        for (int g = 0; g < 3; ++g) { oa[m][g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; oa[m][g][1] = oa[m][g][0]; }
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int il = 0; il < 2; ++il) {
                                const int off = (il * 16 + r) * LDP + 32 * ks + 8 * q;
                                const bf16x8 fh = ldfrag(xh + off), fl = ldfrag(xl + off);
#pragma unroll
                                for (int g = 0; g < 3; ++g) mma_x3(oa[m][g][il], wr_hi[m][ks][g], wr_lo[m][ks][g], fh, fl);
                            }
                    }
                }
                STAMP(0);
#pragma unroll
                for (int il = 0; il < 2; ++il) {
                    LANE_IDS3
                    const int row = 16 * il + r;
                    const f32x4 sr = oa[0][0][il] + oa[1][0][il], sz = oa[0][1][il] + oa[1][1][il];
                    const f32x4 pn = oa[0][2][il], hn = oa[1][2][il];
                    const float deg = sv.deg[row];
                    const float* xt = sv.xtab + sv.cls[row] * 3 * H + c0;
                    const bf16x4 hh = *reinterpret_cast<const bf16x4*>(xp + 2 * PE + row * LDP + c0);
                    const bf16x4 hl = *reinterpret_cast<const bf16x4*>(xp + 3 * PE + row * LDP + c0);
                    const f32x4 bcr = ldv4(sv.bc + c0), bcz = ldv4(sv.bc + H + c0), bcn = ldv4(sv.bc + 2 * H + c0);
                    const f32x4 xr = ldv4(xt), xz = ldv4(xt + H), xn = ldv4(xt + 2 * H), bhn = ldv4(sv.bhh + 2 * H + c0);
                    const f32x4 rr = sigmoid4(sr + (deg * bcr + xr));
                    const f32x4 zz = sigmoid4(sz + (deg * bcz + xz));
                    const f32x4 ghn = hn + bhn;
                    const f32x4 nn = tanh4(pn + (deg * bcn + xn) + rr * ghn);
                    const f32x4 hp = f32x4{(float)hh[0], (float)hh[1], (float)hh[2], (float)hh[3]} + f32x4{(float)hl[0], (float)hl[1], (float)hl[2], (float)hl[3]};
                    const f32x4 pre = nn + zz * (hp - nn);
                    const f32x4 dy = ldv4(s_dy + row * LDF + c0);
                    vr[il] = rr; vz[il] = zz; vn[il] = nn; vg[il] = ghn; vh[il] = hp; vdy[il] = dy;
                    const float s1 = (pre[0] + pre[1]) + (pre[2] + pre[3]);
                    if (has_ln) {
                        const float mean_w = quad_rows_sum3(s1) * (1.0f / 16.0f);
                        const f32x4 gm = ldv4(sv.lnw + c0);
                        const f32x4 d = pre - mean_w, gg = dy * gm;
                        const f32x4 dd = d * d, gd = gg * d;
                        vd[il] = d;
                        const float m2 = quad_rows_sum3((dd[0] + dd[1]) + (dd[2] + dd[3]));
                        const float s3 = quad_rows_sum3((gg[0] + gg[1]) + (gg[2] + gg[3]));
                        const float s4 = quad_rows_sum3((gd[0] + gd[1]) + (gd[2] + gd[3]));
                        mw[il] = mean_w;
                        if (q == 0) s_part[row * 4 + wc] = f32x4{mean_w, m2, s3, s4};
                    } else {
                        vd[il] = pre; mw[il] = 0.f;
                    }
                }
            }
            STAMP(1);
            lds_barrier();                                      // (A) LayerNorm partials of tile p
            STAMP(2);
            // ---- interval 2: P3, LayerNorm + GRU backward; gate gradients to the planes of buffer xb, dh*z to the back waves;
            //      the Whh fragments of tile p-1's dgrad are requested first and used behind P3
            DgW3 wdf;
            const bool fdg = need_dgrad && tB < ntiles;
            if (fdg) {
                int oz = 0;
                asm volatile("" : "+s"(oz));                    // opaque per tile: keeps the (loop-invariant) loads inside the loop
                dgrad_load3<0, 3>(wdf, a.wpack + 6 * BLK + wc * 6 * 512 + oz, lane0);       // k-steps 0-2 now (24 VGPRs beside P3's values)
            }
            if (act) {
                float lnw_acc[4] = {0.f, 0.f, 0.f, 0.f}, lnb_acc[4] = {0.f, 0.f, 0.f, 0.f};
                __bf16* dgp = s_dg + xb * 8 * PE;
#pragma unroll
                for (int il = 0; il < 2; ++il) {
                    LANE_IDS3
                    const int row = 16 * il + r;
                    float dh[4];
                    if (has_ln) {
                        const f32x4 p0 = s_part[row * 4 + 0], p1 = s_part[row * 4 + 1], p2 = s_part[row * 4 + 2], p3 = s_part[row * 4 + 3];
                        const float mean = (p0[0] + p1[0] + p2[0] + p3[0]) * 0.25f;
                        const float e0 = p0[0] - mean, e1 = p1[0] - mean, e2 = p2[0] - mean, e3 = p3[0] - mean;
                        const float var = (p0[1] + p1[1] + p2[1] + p3[1] + 16.0f * (e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3)) * (1.0f / H);
                        const float rstd = rsqrtf(var + a.eps);
                        const float c1 = (p0[2] + p1[2] + p2[2] + p3[2]) * (1.0f / H);
                        const float c2 = (p0[3] + p1[3] + p2[3] + p3[3] + e0 * p0[2] + e1 * p1[2] + e2 * p2[2] + e3 * p3[2]) * rstd * (1.0f / H);
                        const float4 gm = ld4(sv.lnw + c0);
                        const float gm_[4] = {gm.x, gm.y, gm.z, gm.w};
                        const float shift = mw[il] - mean;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float xhat = (vd[il][e] + shift) * rstd;
                            lnw_acc[e] += vdy[il][e] * xhat; lnb_acc[e] += vdy[il][e];
                            dh[e] = rstd * (vdy[il][e] * gm_[e] - c1 - xhat * c2);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) dh[e] = vdy[il][e];
                    }
                    float dar[4], daz[4], dan[4], danr[4];
                    f32x4 dhz;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float rr = vr[il][e], zz = vz[il][e], nn = vn[il][e], ghn = vg[il][e];
                        const float hp = vh[il][e];
                        dan[e] = dh[e] * (1.0f - zz) * (1.0f - nn * nn);
                        daz[e] = dh[e] * (hp - nn) * zz * (1.0f - zz);
                        dar[e] = dan[e] * ghn * rr * (1.0f - rr);
                        danr[e] = dan[e] * rr;
                        dhz[e] = dh[e] * zz;
                    }
                    bf16x4 hi, lo;
                    __bf16* dst = dgp + row * LDP + c0;
                    split4(make_float4(dar[0], dar[1], dar[2], dar[3]), hi, lo);
                    st_bf4(dst, hi); st_bf4(dst + PE, lo);
                    split4(make_float4(daz[0], daz[1], daz[2], daz[3]), hi, lo);
                    st_bf4(dst + 2 * PE, hi); st_bf4(dst + 3 * PE, lo);
                    split4(make_float4(dan[0], dan[1], dan[2], dan[3]), hi, lo);
                    st_bf4(dst + 4 * PE, hi); st_bf4(dst + 5 * PE, lo);
                    split4(make_float4(danr[0], danr[1], danr[2], danr[3]), hi, lo);
                    st_bf4(dst + 6 * PE, hi); st_bf4(dst + 7 * PE, lo);
                    s_dhz[xb * 512 + (wc * 2 + il) * 64 + lane] = dhz;
                }
                if (has_ln) {
                    LANE_IDS3
                    float lw_[4], lb_[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { lw_[e] = group_sum<16>(lnw_acc[e]); lb_[e] = group_sum<16>(lnb_acc[e]); }
                    if (r == 0) {
                        float* acc = s_lnacc + wc * 32 + 4 * q;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { acc[e] += lw_[e]; acc[16 + e] += lb_[e]; }
                    }
                }
            }
            if ((pf0 ^ pf1 ^ pf2) == 0x7fc12345u && a.stamps) a.stamps[0] = pf0;     // keeps the prefetch loads alive; never true in practice
            STAMP(3);
            // ---- the Whh half of tile p-1's dgrad (the back waves run the Wc half meanwhile): d h_in = dh*z + Whh^T [dr, dz, d(n*r)]
            if (fdg) {
                LANE_IDS3
                {
                    int oz = 0;
                    asm volatile("" : "+s"(oz));
                    dgrad_load3<3, 6>(wdf, a.wpack + 6 * BLK + wc * 6 * 512 + oz, lane);           // k-steps 3-5: P3's values are dead, used six steps from here
                }
                const int yb = (p + 1) & 1;
                const __bf16* dgq = s_dg + yb * 8 * PE;
                f32x4 dgo[2] = {s_dhz[yb * 512 + (wc * 2 + 0) * 64 + lane], s_dhz[yb * 512 + (wc * 2 + 1) * 64 + lane]};
                dgrad_mul3(wdf, dgq, dgq + 3 * 2 * PE, dgo, a.g_direct_out, tB * TR, a.N, r, q, c0);
            }
            STAMP(5);
            lds_barrier();                                      // (B) gate-gradient planes and dh*z of tile p
            STAMP(4);
        }
        STAMP_FLUSH(a);
    } else {
        // =================================================================================== BACK waves 4-7
        f32x4 gW[2][3][4];                      // 2x2 block of output tiles per matrix and gate (weight gradients)
        f32x4 gX[2][2];                         // bias-type gradients: planes p = m (pp 0) and 2 + m (pp 1), gate-column tile wc
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t) gW[m][g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            gX[m][0] = gX[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        int s0 = 0, s1 = 1;                       // ring slots of tile p (free: refilled with tile p + 3) and of tile p + 1
        STAMP_DECL
        for (int p = 0;; ++p, s0 = s1, s1 = s1 == 2 ? 0 : s1 + 1) {
            const int64_t tF = seq.at(p), tB = p >= 1 ? seq.at(p - 1) : ntiles, tN = seq.at(p + 1);
            if (tF >= ntiles && tB >= ntiles) break;
            STAMP_BEGIN;
            const bool actB = tB < ntiles, actN = tF < ntiles && tN < ntiles;
            const int yb = (p + 1) & 1;                         // buffers of tile p-1 (and of tile p+1)
            const __bf16* xp = s_x + yb * 4 * PE;
            const __bf16* dgp = s_dg + yb * 8 * PE;
            // ---- interval 1: weight gradients of tile p-1, then the row gather of tile p+1 (registers), pointers of tile p+3.
            //      (Requesting the rows BEFORE the weight gradients, so that they fly meanwhile, was tried: rows3_issue / rows3_finish
            //      exist for it; with the 112 accumulators the in-flight rows do not fit the register file: 56-77 spilled registers.)
            if (actB) {
                // Weight gradients of both matrices over the tile's 32 rows (one k-step): the transposed fragments of the wave's two
                // input-column tiles (and of the [deg, onehot, 1] columns) serve all three gates; the bias-type tile (gX) reuses the
                // gate-gradient fragment the 2x2 block loads anyway.
                const int it0 = 2 * (wc >> 1), jt0 = 2 * (wc & 1), ig = wc & 1;
                const __bf16* xe_hi = s_xe + yb * 2 * TR * XLD;
                const __bf16* xe_lo = xe_hi + TR * XLD;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const __bf16* x_hi = xp + (2 * m) * PE;
                    const __bf16* x_lo = x_hi + PE;
                    bf16x8 bh[2], bl[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) { bh[j] = ldfrag_tr(x_hi, LDP, 0, (jt0 + j) * 16); bl[j] = ldfrag_tr(x_lo, LDP, 0, (jt0 + j) * 16); }
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const int pl = g == 2 ? 2 + m : g;
                        const __bf16* ph = dgp + pl * 2 * PE;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const bf16x8 ah = ldfrag_tr(ph, LDP, 0, (it0 + i) * 16), al = ldfrag_tr(ph + PE, LDP, 0, (it0 + i) * 16);
#pragma unroll
                            for (int j = 0; j < 2; ++j) mma_x3(gW[m][g][i * 2 + j], ah, al, bh[j], bl[j]);
                            if (i == ig && (g == 2 || g == m)) mma_x3(gX[m][g == 2 ? 1 : 0], ah, al, ldfrag_tr(xe_hi, XLD, 0, 0), ldfrag_tr(xe_lo, XLD, 0, 0));
                        }
                    }
                }
            }
            STAMP(0);
            RowPack pk;
            if (actN) {
                int t_ = tid; asm volatile("" : "+v"(t_));
                const int grp = (t_ - 256) >> 4, lr = t_ & 15;
                float4 acc[2], own[2], dy[2];
                float deg[2];
                int cls[2];
                tile_rows3(a, tN * TR, grp, lr, idxl3(idx_base, s1).ptr, idxl3(idx_base, s1).idx, *idxl3(idx_base, s1).dmax(), acc, own, dy, deg, cls);
                prep_rows(pk, acc, own, dy, deg, cls);
            }
            if (bt <= TR) idxl3(idx_base, s0).ptr[bt] = rp;         // tile p+3 into the ring slot of tile p (its gather ended a period ago)
            STAMP(1);
            lds_barrier();                                      // (A) operand planes of tile p-1 are free
            STAMP(2);
            // ---- interval 2: operand planes of tile p+1, indices of tile p+3, dgrad of tile p-1
            DgW3 wdb;
            if (actB && need_dgrad) {
                int oz = 0;
                asm volatile("" : "+s"(oz));                    // opaque per tile: keeps the (loop-invariant) loads inside the loop
                dgrad_load3<0, 6>(wdb, a.wpack + 4 * BLK + wc * 6 * 512 + oz, lane0);
            }
            if (actN) store_rows(yb, pk);
            {
                const int* sp = idxl3(idx_base, s0).ptr;        // tile p+3's pointers (written in interval 1)
                const int e0 = sp[0], ne = sp[TR] - e0;
                ri = bt < ne ? a.idx[e0 + bt] : 0;
                if (w == 4) {
                    int d = lane0 < TR ? sp[lane0 + 1] - sp[lane0] : 0;
#pragma unroll
                    for (int mm = 32; mm >= 1; mm >>= 1) d = max(d, __shfl_xor(d, mm, 64));
                    if (lane0 == 0) *idxl3(idx_base, s0).dmax() = ne > B3::IDXCAP - 8 ? (1 << 30) : d;
                }
                rp = ptr_prefetch3(a, seq.at(p + 4), ntiles, bt);
            }
            STAMP(3);
            if (actB && need_dgrad) {
                // the Wc half of tile p-1's dgrad: d agg = Wc^T [dr, dz, dn] (the front waves run the Whh half behind their P3)
                LANE_IDS3
                f32x4 dgo[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                dgrad_mul3(wdb, dgp, dgp + 2 * 2 * PE, dgo, a.g_agg_out, tB * TR, a.N, r, q, c0);
            }
            idxl3(idx_base, s0).idx[bt] = ri;                   // tile p+3's indices (requested at the head of this interval)
            STAMP(4);
            lds_barrier();                                      // (B) operand planes of tile p+1, index ring slot of tile p+3
            STAMP(5);
        }
        STAMP_FLUSH(a);
        // ---- flush: per-workgroup slab, lane-linear float4 slots (k_struct_stage_bwd2_reduce's mapping, block m * 4 + wc)
        float* slab = args.slab + (int64_t)blockIdx.x * B3::SLAB;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4* sw = reinterpret_cast<f32x4*>(slab) + ((m * 4 + wc) * B3::SLOTS) * 64 + lane0;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t) sw[(g * 4 + t) * 64] = gW[m][g][t];
            sw[12 * 64] = gX[m][0];
            sw[13 * 64] = gX[m][1];
        }
    }
    // LayerNorm affine gradients of the four front waves
    __syncthreads();
    if (tid < 2 * H) {
        const int which = tid >> 6, c = tid & 63, wcc = c >> 4, cc = c & 15;
        float* slab = args.slab + (int64_t)blockIdx.x * B3::SLAB;
        slab[B3::SLAB_W + which * H + c] = s_lnacc[wcc * 32 + which * 16 + cc];
    }
}

int launch_bwd3_x3(const StageX3Args& s, float* workspace, int64_t workspace_floats, hipStream_t st) {
    static bool set[64] = {false};      // per device: each device loads its own copy of the code object
    int dev = 0;
    hipGetDevice(&dev);
    if (!set[dev & 63]) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_struct_stage_bwd3_x3), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set[dev & 63] = true; }
    const int64_t ntiles = (s.N + B3::TR - 1) / B3::TR;
    const int grid = grid_for(ntiles, 1);
    if (workspace == nullptr || workspace_floats < (int64_t)grid * B3::SLAB) return MGV_EINVAL;
    B3Args a{s, workspace};
    hipLaunchKernelGGL(k_struct_stage_bwd3_x3, dim3(grid), dim3(kThreadsX3), B3::bytes, st, a);
    return launch_stage_slab_reduce(s, workspace, grid, st);
}

}  // namespace mgv

#ifdef MGV_STAMPS
static unsigned long long* g_stamps3 = nullptr;
extern "C" int mgv_diag_set_stamps3(void* p) { g_stamps3 = static_cast<unsigned long long*>(p); return 0; }
#define MGV_SET_STAMPS3(a) (a).stamps = g_stamps3
#else
#define MGV_SET_STAMPS3(a)
#endif

extern "C" int mgv_struct_stage_bwd3_x3(int H, int64_t N, const float* h_in, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                                        const uint8_t* xcls, const float* xtab, int C, const void* wpack_bf16, const float* bc,
                                        const float* bhh, const float* ln_w, const float* ln_b, float ln_eps,
                                        const float* gy_direct, const float* gy_agg, float* g_direct_out, float* g_agg_out,
                                        float* dWc, float* dbc, float* dWhh, float* dbhh, float* dxtab, float* dln_w,
                                        float* dln_b, float* workspace, int64_t workspace_floats, int heavy_n,
                                        const int32_t* heavy_nodes, float* heavy_ws, const int32_t* table_own_idx, void* stream) {
    MGV_CHECK_ARG(N >= 0 && h_in && nbr_ptr && xcls && xtab && wpack_bf16 && bc && bhh && gy_direct);
    MGV_CHECK_ARG(dWc && dbc && dWhh && dbhh && dxtab);
    MGV_CHECK_ARG(C >= 1 && C <= mgv::kMaxClsX3);
    MGV_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr));
    MGV_CHECK_ARG(ln_w == nullptr || (dln_w && dln_b));
    MGV_CHECK_ARG((g_direct_out == nullptr) == (g_agg_out == nullptr));
    if (H != 64) return MGV_EUNSUPPORTED;
    if (N == 0) return MGV_OK;
    MGV_CHECK_ARG(nbr_idx != nullptr);
    mgv::StageX3Args a{};
    a.N = N; a.h_in = h_in; a.ptr = nbr_ptr; a.idx = nbr_idx; a.xcls = xcls; a.xtab = xtab; a.C = C;
    a.wpack = static_cast<const __bf16*>(wpack_bf16); a.bc = bc; a.bhh = bhh; a.lnw = ln_w; a.lnb = ln_b; a.eps = ln_eps;
    a.gy_direct = gy_direct; a.gy_agg = gy_agg; a.g_direct_out = g_direct_out; a.g_agg_out = g_agg_out;
    a.dWc = dWc; a.dbc = dbc; a.dWhh = dWhh; a.dbhh = dbhh; a.dxtab = dxtab; a.dlnw = dln_w; a.dlnb = dln_b;
    a.gmask = -1;
    MGV_CHECK_ARG(table_own_idx == nullptr || N < (1 << 24));
    if (table_own_idx) { a.hshift = 24; a.gmask = 0xffffff; a.own_idx = table_own_idx; }
    MGV_SET_STAMPS3(a);
    { static const int v = [] { const char* e = getenv("MGV_XCD_TILES"); return (e && e[0] == '0') ? 0 : 1; }(); a.xcd = v; }
    { static const int v = [] { const char* e = getenv("MGV_ROW_PREFETCH"); return (e && e[0] == '0') ? 0 : 1; }(); a.prefetch = v; }
    MGV_CHECK_ARG(heavy_n >= 0 && (heavy_n == 0 || (heavy_nodes && heavy_ws)));
    mgv::launch_heavy_sums<64>(a, heavy_n, heavy_nodes, heavy_ws, gy_agg != nullptr, static_cast<hipStream_t>(stream));
    return mgv::launch_bwd3_x3(a, workspace, workspace_floats, static_cast<hipStream_t>(stream));
}
