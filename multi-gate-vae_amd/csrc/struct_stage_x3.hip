// Structural-encoder half round on bf16x3 split-precision MFMA (see mgv_x3.h); same math, arguments
// and phases as struct_stage.hip, whose header describes the operator.  With the dense part 5x
// cheaper the kernel is bound by the row gathers: HBM bytes per node and half round are unchanged
// (forward (deg+2)*4H, backward (2deg+4)*4H).
//
// Weights arrive pre-split by the host in one bf16 pack of eight [3H*H] blocks:
//   0 Wc_hi  1 Wc_lo  2 Whh_hi  3 Whh_lo   ([3H][H], k contiguous: forward B operands)
//   4 WcT_hi 5 WcT_lo 6 WhhT_hi 7 WhhT_lo  ([H][3H], k contiguous: dgrad B operands)
#include "struct_stage_x3_common.h"

#ifndef MGV_FWD_D
#define MGV_FWD_D 3            // neighbour slots per row and gather round of the H = 64 forward
#endif

namespace mgv {

// gate pre-activations of one tile from the split planes; weights streamed from L2
template <int H>
__device__ __forceinline__ void stage_gemm_x3(const __bf16* wpack, const __bf16* wc_hi, const __bf16* whh_hi, const __bf16* agg_hi, const __bf16* agg_lo,
                                              const __bf16* hin_hi, const __bf16* hin_lo,
                                              f32x4 (&ar)[SplitX3<H>::RTW][SplitX3<H>::HCW],
                                              f32x4 (&az)[SplitX3<H>::RTW][SplitX3<H>::HCW],
                                              f32x4 (&ani)[SplitX3<H>::RTW][SplitX3<H>::HCW],
                                              f32x4 (&anh)[SplitX3<H>::RTW][SplitX3<H>::HCW]) {
    using S = SplitX3<H>;
    constexpr int LDP = H + 8, BLK = 3 * H * H;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
#pragma unroll
    for (int i = 0; i < S::RTW; ++i)
#pragma unroll
        for (int j = 0; j < S::HCW; ++j) {
            ar[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; az[i][j] = ar[i][j]; ani[i][j] = ar[i][j]; anh[i][j] = ar[i][j];
        }
#pragma unroll 1
    for (int ks = 0; ks < H / 32; ++ks) {
        const int ko = 32 * ks + 8 * q;
        bf16x8 ah[S::RTW], al[S::RTW], hh[S::RTW], hl[S::RTW];
#pragma unroll
        for (int i = 0; i < S::RTW; ++i) {
            const int off = ((wr * S::RTW + i) * 16 + r) * LDP + ko;
            ah[i] = ldfrag(agg_hi + off); al[i] = ldfrag(agg_lo + off);
            hh[i] = ldfrag(hin_hi + off); hl[i] = ldfrag(hin_lo + off);
        }
#pragma unroll
        for (int j = 0; j < S::HCW; ++j) {
            const int col = (wc * S::HCW + j) * 16 + r;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                // fragment order: block (row tile, k-step) = 512 contiguous elements, lane l at 8*l
                const int wo = ((g * (H / 16) + wc * S::HCW + j) * (H / 32) + ks) * 512 + lane * 8;
                const int lo = (g * H + col) * (H + 8) + ko;           // row-major hi planes in LDS
                const bf16x8 ch = ldfrag(wc_hi + lo), cl = ldfrag(wpack + 1 * BLK + wo);
                const bf16x8 uh = ldfrag(whh_hi + lo), ul = ldfrag(wpack + 3 * BLK + wo);
#pragma unroll
                for (int i = 0; i < S::RTW; ++i) {
                    if (g == 0) { mma_x3(ar[i][j], ah[i], al[i], ch, cl); mma_x3(ar[i][j], hh[i], hl[i], uh, ul); }
                    if (g == 1) { mma_x3(az[i][j], ah[i], al[i], ch, cl); mma_x3(az[i][j], hh[i], hl[i], uh, ul); }
                    if (g == 2) { mma_x3(ani[i][j], ah[i], al[i], ch, cl); mma_x3(anh[i][j], hh[i], hl[i], uh, ul); }
                }
                __builtin_amdgcn_sched_barrier(0);      // keep one gate's weight fragments live at a time
            }
        }
    }
}

// Forward kernel: 4 waves per workgroup, two workgroups per CU.  Every wave owns one 16-column tile of
// the three gates for ALL its tiles, so its 2*3*(H/32) weight fragment pairs (96 VGPRs at H=64) are
// loaded once per kernel and stay in registers: the dense phase touches LDS only.
constexpr int kThreadsF = 256;

#ifndef MGV_ABLF
#define MGV_ABLF 0           // timing ablations of diagnostic builds of the FORWARD kernel (results are wrong): 1 no MFMA, 2 light epilogue, 4 no row gathers, 8 no output stores
#endif
#if MGV_ABLF & 1
#define FWD_MMA(c, ah, al, bh, bl) asm volatile("" :: "v"(ah), "v"(al), "v"(bh), "v"(bl))
#else
#define FWD_MMA(c, ah, al, bh, bl) mma_x3(c, ah, al, bh, bl)
#endif
template <int H>
__global__ __launch_bounds__(kThreadsF, 2) void k_struct_stage_fwd_x3(StageX3Args a) {
    using S = WaveSplit<H>;                 // 4 waves: column tiles first, then row tiles (mgv_common.h)
    using M = X3Smem<H>;
    static_assert(S::HCW == 1, "one hidden-column tile per wave");
    constexpr int LDP = M::LDP, BLK = 3 * H * H, KS = H / 32;
    constexpr int RPG = kTileRows / S::GROUPS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* agg_hi = reinterpret_cast<__bf16*>(smem_raw + M::f_planes);
    __bf16* agg_lo = reinterpret_cast<__bf16*>(smem_raw + M::f_planes + M::PB);
    __bf16* hin_hi = reinterpret_cast<__bf16*>(smem_raw + M::f_planes + 2 * M::PB);
    __bf16* hin_lo = reinterpret_cast<__bf16*>(smem_raw + M::f_planes + 3 * M::PB);
    float* s_hin = reinterpret_cast<float*>(smem_raw + M::f_hin);
    float* s_pre = reinterpret_cast<float*>(smem_raw + M::f_pre);
    const SmallVecs sv = stage_small<H>(a, reinterpret_cast<float*>(smem_raw + M::f_small));
    fold_bhh_rz<H>(a, sv);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
    const int grp = tid / S::LPR, lr = tid % S::LPR;
    const bool has_ln = a.lnw != nullptr;
    const int64_t ntiles = (a.N + kTileRows - 1) / kTileRows;

    // persistent weight fragments of this wave's column tile: [k-step][gate] x {Wc hi, Wc lo, Whh hi, Whh lo}
    bf16x8 wch[KS][3], wcl[KS][3], wuh[KS][3], wul[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int wo = ((g * (H / 16) + wc) * KS + ks) * 512 + lane * 8;
            wch[ks][g] = ldfrag(a.wpack + 0 * BLK + wo); wcl[ks][g] = ldfrag(a.wpack + 1 * BLK + wo);
            wuh[ks][g] = ldfrag(a.wpack + 2 * BLK + wo); wul[ks][g] = ldfrag(a.wpack + 3 * BLK + wo);
        }

    // Index pipeline, two LDS buffers: while tile t runs, buffer b holds the CSR pointers / neighbour ids of tile t, buffer b^1
    // those of tile t+1 (committed at the end of tile t-1); the ids of tile t+2 are requested after tile t's dense part and committed
    // into buffer b at its end, a whole tile before their row phase.  (An L2 prefetch of tile t+1's rows from here was measured: +8 %
    // time, the kernel sits at the CU's memory-instruction rate and every extra load costs; the backward kernel, which idles longer, keeps it.)
    int* idx_base = reinterpret_cast<int*>(smem_raw + M::f_idx);
    const TileSeq seq = tile_seq(ntiles, a.xcd);
    int ri[kIdxCap / kThreadsF];
    int rp;
    for (int k = 0; k < 2; ++k) {
        rp = ptr_prefetch(a, seq.at(k), ntiles);
        if (tid <= kTileRows) idx_lds(idx_base, k).ptr[tid] = rp;
        __syncthreads();
        idx_prefetch<kThreadsF>(a, idx_lds(idx_base, k).ptr, ri);
        idx_commit<kThreadsF>(idx_lds(idx_base, k).idx, ri);
        tile_dmax(idx_lds(idx_base, k).ptr, idx_lds(idx_base, k).dmax());
    }
    rp = ptr_prefetch(a, seq.at(2), ntiles);
    __syncthreads();
    int b = 0;
    STAMP_DECL
    for (int it = 0; seq.at(it) < ntiles; ++it, b ^= 1) {
        const int64_t tile = seq.at(it);
        const int64_t base = tile * kTileRows;
        STAMP_BEGIN;
        // ---- row phase: independent loads only
        {
            float4 acc[RPG], own[RPG], dy[RPG];
            float deg[RPG];
            int cls[RPG];
#if MGV_ABLF & 4
            for (int rr = 0; rr < RPG; ++rr) { acc[rr] = make_float4(0.1f * lr, 0.2f, 0.3f, 0.4f); own[rr] = acc[rr]; deg[rr] = 2.f; cls[rr] = 1; }
#else
            tile_rows<H, RPG, false, (H == 64 ? MGV_FWD_D : 4)>(a, base, grp, S::GROUPS, lr, idx_lds(idx_base, b).ptr, idx_lds(idx_base, b).idx, *idx_lds(idx_base, b).dmax(), acc, own, dy, deg, cls);
#endif
#pragma unroll
            for (int rr = 0; rr < RPG; ++rr) {
                const int row = grp + rr * S::GROUPS;
                bf16x4 hi, lo;
                split4(acc[rr], hi, lo);
                st_bf4(agg_hi + row * LDP + 4 * lr, hi); st_bf4(agg_lo + row * LDP + 4 * lr, lo);
                split4(own[rr], hi, lo);
                st_bf4(hin_hi + row * LDP + 4 * lr, hi); st_bf4(hin_lo + row * LDP + 4 * lr, lo);
                st4(s_hin + row * S::LD + 4 * lr, own[rr]);
                if (lr == 0) { sv.deg[row] = deg[rr]; sv.cls[row] = cls[rr]; }
            }
        }
        STAMP(0);
        __syncthreads();
        STAMP(1);
        if (tid <= kTileRows) idx_lds(idx_base, b).ptr[tid] = rp;           // this tile's index buffer is dead: pointers of tile t+2 (requested one tile ago)
        // ---- dense part: LDS fragments x register-resident weights
        f32x4 ar[S::RTW], az[S::RTW], ani[S::RTW], anh[S::RTW];
#pragma unroll
        for (int i = 0; i < S::RTW; ++i) { ar[i] = f32x4{0.f, 0.f, 0.f, 0.f}; az[i] = ar[i]; ani[i] = ar[i]; anh[i] = ar[i]; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < S::RTW; ++i) {
                const int off = ((wr * S::RTW + i) * 16 + r) * LDP + 32 * ks + 8 * q;
                const bf16x8 ah = ldfrag(agg_hi + off), al = ldfrag(agg_lo + off);
                const bf16x8 hh = ldfrag(hin_hi + off), hl = ldfrag(hin_lo + off);
                FWD_MMA(ar[i], ah, al, wch[ks][0], wcl[ks][0]); FWD_MMA(ar[i], hh, hl, wuh[ks][0], wul[ks][0]);
                FWD_MMA(az[i], ah, al, wch[ks][1], wcl[ks][1]); FWD_MMA(az[i], hh, hl, wuh[ks][1], wul[ks][1]);
                FWD_MMA(ani[i], ah, al, wch[ks][2], wcl[ks][2]); FWD_MMA(anh[i], hh, hl, wuh[ks][2], wul[ks][2]);
            }
        STAMP(2);
        __syncthreads();        // s_pre overlays the agg planes; pointers of tile t+2 are in buffer b
        STAMP(3);
        // ids of tile t+2 and pointers of tile t+3 fly during the epilogue
        idx_prefetch<kThreadsF>(a, idx_lds(idx_base, b).ptr, ri);
        rp = ptr_prefetch(a, seq.at(it + 3), ntiles);
        tile_dmax(idx_lds(idx_base, b).ptr, idx_lds(idx_base, b).dmax());
        {   // GRU on 4-vectors (a lane's accumulator = four consecutive rows of one column): the arithmetic around the
            // transcendentals compiles to packed fp32 operations; b_hr, b_hz sit in the LDS class table (fold_bhh_rz)
            const int col = wc * 16 + r;
            const float bcr = sv.bc[col], bcz = sv.bc[H + col], bcn = sv.bc[2 * H + col], bhn = sv.bhh[2 * H + col];
#pragma unroll
            for (int i = 0; i < S::RTW; ++i) {
                const int row0 = (wr * S::RTW + i) * 16 + q * 4;
                const f32x4 deg = ldv4(sv.deg + row0);
                const int4 cls = *reinterpret_cast<const int4*>(sv.cls + row0);
                const float* x0 = sv.xtab + cls.x * 3 * H + col;
                const float* x1 = sv.xtab + cls.y * 3 * H + col;
                const float* x2 = sv.xtab + cls.z * 3 * H + col;
                const float* x3 = sv.xtab + cls.w * 3 * H + col;
                const f32x4 xr = f32x4{x0[0], x1[0], x2[0], x3[0]}, xz = f32x4{x0[H], x1[H], x2[H], x3[H]}, xn = f32x4{x0[2 * H], x1[2 * H], x2[2 * H], x3[2 * H]};
                const float* hrow = s_hin + row0 * S::LD + col;
                const f32x4 hp = f32x4{hrow[0], hrow[S::LD], hrow[2 * S::LD], hrow[3 * S::LD]};
#if MGV_ABLF & 2
                const f32x4 pre = ar[i] + az[i] + ani[i] + anh[i] + hp + xr + xz + xn + deg * (bcr + bcz + bcn + bhn);
#else
                const f32x4 rr = sigmoid4(ar[i] + (deg * bcr + xr));
                const f32x4 zz = sigmoid4(az[i] + (deg * bcz + xz));
                const f32x4 nn = tanh4(ani[i] + (deg * bcn + xn) + rr * (anh[i] + bhn));
                const f32x4 pre = nn + zz * (hp - nn);
#endif
                float* prow = s_pre + row0 * S::LD + col;
                prow[0] = pre[0]; prow[S::LD] = pre[1]; prow[2 * S::LD] = pre[2]; prow[3 * S::LD] = pre[3];
            }
        }
        STAMP(4);
        STAMP(5);
        __syncthreads();
        STAMP(6);
        for (int row = grp; row < kTileRows; row += S::GROUPS) {
            const int64_t node = base + row;
            float4 v = ld4(s_pre + row * S::LD + 4 * lr);
            if (has_ln) {
                const float mean = group_sum<S::LPR>(v.x + v.y + v.z + v.w) * (1.0f / H);
                v = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
                const float var = group_sum<S::LPR>(dot4(v, v)) * (1.0f / H);
                const float rstd = rsqrtf(var + a.eps);
                if (a.ln_stats && lr < 2 && node < a.N) {      // kept for the backward (bwd2); buffer store: a 32-bit offset, no address registers
                    const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.ln_stats, 0, 0xfffffff0u, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lr ? rstd : mean), rs, (unsigned)node * 8u + 4u * lr, 0, 0);
                }
                const float4 g = ld4(sv.lnw + 4 * lr), bb = ld4(sv.lnb + 4 * lr);
                v = make_float4(v.x * rstd * g.x + bb.x, v.y * rstd * g.y + bb.y, v.z * rstd * g.z + bb.z, v.w * rstd * g.w + bb.w);
            }
#if MGV_ABLF & 8
            if (node < a.N && v.x == 1.2345e-30f) st4(a.h_out + node * H + 4 * lr, v);
#else
            if (node < a.N) st4(a.h_out + node * H + 4 * lr, v);
#endif
        }
        idx_commit<kThreadsF>(idx_lds(idx_base, b).idx, ri);               // ids of tile t+2 replace this tile's
        STAMP(7);
        __syncthreads();
    }
    STAMP_FLUSH(a);
}

// ------------------------------------------------------------------------------------------ backward
template <int H>
struct WgradX3 {
    static constexpr int HC = H / 16;
    static constexpr int T = HC * HC;
    static_assert(T >= 4, "bf16x3 path needs H >= 32");
    static constexpr bool KSPLIT = T < kNW;                 // fewer tiles than waves: waves share a tile, split the 64 rows
    static constexpr int TPW = KSPLIT ? 1 : T / kNW;
    static constexpr int KS = KSPLIT ? (kTileRows / 32) / (kNW / T) : kTileRows / 32;   // 32-row k-steps per wave
    static_assert(KS >= 1, "k split too fine");
    __device__ static int tile_of(int w, int t) { return KSPLIT ? w % T : w * TPW + t; }
    __device__ static int ks0(int w) { return KSPLIT ? (w / T) * KS : 0; }
};

// acc[t] += D^T[gate cols of tile][64 rows] * X[64 rows][input cols of tile]; both operands are read
// transposed (ds_read_b64_tr_b16) from the row-major planes the forward-orientation products also use
template <int H>
__device__ __forceinline__ void wgrad_x3(f32x4 (&acc)[WgradX3<H>::TPW], const __bf16* d_hi, const __bf16* d_lo,
                                         const __bf16* x_hi, const __bf16* x_lo) {
    using W = WgradX3<H>;
    constexpr int LDP = H + 8;
    const int w = threadIdx.x >> 6;
#pragma unroll 1
    for (int kk = 0; kk < W::KS; ++kk) {
        const int k0 = 32 * (W::ks0(w) + kk);
#pragma unroll
        for (int t = 0; t < W::TPW; ++t) {
            const int tl = W::tile_of(w, t);
            const int it = tl / W::HC, jt = tl % W::HC;
            const bf16x8 ah = ldfrag_tr2(d_hi, LDP, k0, it * 16), al = ldfrag_tr2(d_lo, LDP, k0, it * 16);
            const bf16x8 bh = ldfrag_tr2(x_hi, LDP, k0, jt * 16), bl = ldfrag_tr2(x_lo, LDP, k0, jt * 16);
            mma_x3(acc[t], ah, al, bh, bl);
        }
    }
}

template <int H>
__device__ __forceinline__ void wgrad_flush_x3(const f32x4 (&acc)[WgradX3<H>::TPW], float* dW) {
    using W = WgradX3<H>;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t = 0; t < W::TPW; ++t) {
        const int tl = W::tile_of(w, t);
        const int it = tl / W::HC, jt = tl % W::HC;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dW + (int64_t)(it * 16 + q * 4 + e) * H + jt * 16 + r, acc[t][e]);
    }
}

__device__ __forceinline__ void colsum_lds_x3(float v, float* dst) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if ((threadIdx.x & 63) < 16) atomicAdd(dst, v);
}

template <int H>
__global__ __launch_bounds__(kThreadsX3) void k_struct_stage_bwd_x3(StageX3Args a) {
    using S = SplitX3<H>;
    using M = X3Smem<H>;
    using W = WgradX3<H>;
    constexpr int LDP = M::LDP, BLK = 3 * H * H;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* agg_hi = reinterpret_cast<__bf16*>(smem_raw + M::b_planes);
    __bf16* agg_lo = reinterpret_cast<__bf16*>(smem_raw + M::b_planes + M::PB);
    __bf16* hin_hi = reinterpret_cast<__bf16*>(smem_raw + M::b_planes + 2 * M::PB);
    __bf16* hin_lo = reinterpret_cast<__bf16*>(smem_raw + M::b_planes + 3 * M::PB);
    float* s_pre = reinterpret_cast<float*>(smem_raw + M::b_c);
    float* s_dy = reinterpret_cast<float*>(smem_raw + M::b_c + M::F32TILE);
    __bf16* d_hi = reinterpret_cast<__bf16*>(smem_raw + M::b_c);                       // aliases s_pre/s_dy
    const SmallVecs sv = stage_small<H>(a, reinterpret_cast<float*>(smem_raw + M::b_small));
    __bf16* wc_hi = reinterpret_cast<__bf16*>(smem_raw + M::b_w);
    __bf16* whh_hi = reinterpret_cast<__bf16*>(smem_raw + M::b_w + M::WPB);
    {   // fragment-order hi blocks of the pack (0: Wc, 2: Whh) -> row-major LDS planes, 8 k-contiguous elements per copy
        constexpr int KSN = H / 32;
        for (int v = threadIdx.x; v < 3 * H * H / 8; v += kThreadsX3) {
            const int blk = v >> 6, ln = v & 63;
            const int row = (blk / KSN) * 16 + (ln & 15), k = (blk % KSN) * 32 + (ln >> 4) * 8;
            *reinterpret_cast<bf16x8*>(wc_hi + row * M::WLD + k) = ldfrag(a.wpack + 0 * 3 * H * H + v * 8);
            *reinterpret_cast<bf16x8*>(whh_hi + row * M::WLD + k) = ldfrag(a.wpack + 2 * 3 * H * H + v * 8);
        }
    }
    float* s_stat = reinterpret_cast<float*>(smem_raw + M::b_stat);
    float* s_dxt = reinterpret_cast<float*>(smem_raw + M::b_acc);
    float* s_dbc = s_dxt + kMaxClsX3 * 3 * H;
    float* s_dbhh = s_dbc + 3 * H;
    float* s_dlnw = s_dbhh + 3 * H;
    float* s_dlnb = s_dlnw + H;
    __bf16* xe_hi = reinterpret_cast<__bf16*>(smem_raw + M::b_xe);
    __bf16* xe_lo = xe_hi + kTileRows * XLD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    for (int i = tid; i < M::ACC_F; i += kThreadsX3) s_dxt[i] = 0.f;
    const int wc = w % S::WPC, wr = w / S::WPC;
    const int grp = tid / S::LPR, lr = tid % S::LPR;
    const bool has_ln = a.lnw != nullptr;
    const bool need_dgrad = a.g_direct_out != nullptr;
    const int64_t ntiles = (a.N + kTileRows - 1) / kTileRows;

    // weight-gradient accumulators, persistent over the workgroup's tiles.  WBLK (H = 64): gWc[g] holds the wave's 2x2
    // block of ITS matrix (gWhh unused); otherwise TPW tiles of each matrix per wave.
    constexpr bool WBLK = (H == 64 && kNW == 8 && kTPR == 2);
    constexpr int GT = WBLK ? 4 : W::TPW;
    f32x4 gWc[3][GT], gWhh[3][WBLK ? 1 : W::TPW];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int t = 0; t < GT; ++t) gWc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < (WBLK ? 1 : W::TPW); ++t) gWhh[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // bias-type gradients (dbc, dxtab, dbhh) as one more wgrad tile per pass: dG^T x [deg, onehot(cls), 1]
    f32x4 gX[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) gX[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    int* idx_base = reinterpret_cast<int*>(smem_raw + M::b_idx);
    const TileSeq seq = tile_seq(ntiles, a.xcd);
    int rp = ptr_prefetch(a, seq.at(0), ntiles);
    if (tid <= kTileRows) idx_lds(idx_base, 0).ptr[tid] = rp;
    __syncthreads();
    int ri[kIdxCap / kThreadsX3];
    idx_prefetch<kThreadsX3>(a, idx_lds(idx_base, 0).ptr, ri);
    idx_commit<kThreadsX3>(idx_lds(idx_base, 0).idx, ri);
    tile_dmax(idx_lds(idx_base, 0).ptr, idx_lds(idx_base, 0).dmax());
    rp = ptr_prefetch(a, seq.at(1), ntiles);
    __syncthreads();
    int b = 0;
    STAMP_DECL
    constexpr int RPG = kTileRows / S::GROUPS;
    for (int it = 0; seq.at(it) < ntiles; ++it, b ^= 1) {
        const int64_t tile = seq.at(it);
        const int64_t base = tile * kTileRows;
        STAMP_BEGIN;
        // ---- A. row phase (independent loads); operand planes row-major (GEMM A operands) and
        //        (they serve the wgrad too, read transposed)
        {
            float4 acc[RPG], own[RPG], dy[RPG];
            float deg[RPG];
            int cls[RPG];
            tile_rows<H, RPG, true>(a, base, grp, S::GROUPS, lr, idx_lds(idx_base, b).ptr, idx_lds(idx_base, b).idx, *idx_lds(idx_base, b).dmax(), acc, own, dy, deg, cls);
#pragma unroll
            for (int rr = 0; rr < RPG; ++rr) {
                const int row = grp + rr * S::GROUPS;
                bf16x4 hi, lo;
                split4(acc[rr], hi, lo);
                st_bf4(agg_hi + row * LDP + 4 * lr, hi); st_bf4(agg_lo + row * LDP + 4 * lr, lo);
                split4(own[rr], hi, lo);
                st_bf4(hin_hi + row * LDP + 4 * lr, hi); st_bf4(hin_lo + row * LDP + 4 * lr, lo);
                st4(s_dy + row * S::LD + 4 * lr, dy[rr]);
                if (lr == 0) { sv.deg[row] = deg[rr]; sv.cls[row] = cls[rr]; }
                for (int jx = lr; jx < 16; jx += S::LPR) {       // the row group's lanes share the 16 extra columns
                    const float xe = jx == 0 ? deg[rr] : (jx <= 8 ? (cls[rr] == jx - 1 ? 1.0f : 0.0f) : (jx == 9 ? 1.0f : 0.0f));
                    __bf16 xh, xl;
                    split_bf16(xe, xh, xl);
                    xe_hi[row * XLD + jx] = xh; xe_lo[row * XLD + jx] = xl;
                }
            }
        }
        if (tid <= kTileRows) idx_lds(idx_base, b ^ 1).ptr[tid] = rp;
        STAMP(0);
        __syncthreads();
        STAMP(1);
        // ---- B. recompute gates; keep the own-row values (hi+lo) for the GRU backward
        f32x4 ar[S::RTW][S::HCW], az[S::RTW][S::HCW], ani[S::RTW][S::HCW], anh[S::RTW][S::HCW];
        stage_gemm_x3<H>(a.wpack, wc_hi, whh_hi, agg_hi, agg_lo, hin_hi, hin_lo, ar, az, ani, anh);
        idx_prefetch<kThreadsX3>(a, idx_lds(idx_base, b ^ 1).ptr, ri);          // after the weight fragments (vmcnt is in order); committed at the tile's end
        rp = ptr_prefetch(a, seq.at(it + 2), ntiles);
        tile_dmax(idx_lds(idx_base, b ^ 1).ptr, idx_lds(idx_base, b ^ 1).dmax());
        STAMP(2);
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int j = 0; j < S::HCW; ++j) {
                const int col = (wc * S::HCW + j) * 16 + r;
                const float bcr = sv.bc[col], bcz = sv.bc[H + col], bcn = sv.bc[2 * H + col];
                const float bhr = sv.bhh[col], bhz = sv.bhh[H + col], bhn = sv.bhh[2 * H + col];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                    const float deg = sv.deg[row];
                    const float* xt = sv.xtab + sv.cls[row] * 3 * H;
                    const float rr = sigmoidf_(ar[i][j][e] + deg * bcr + xt[col] + bhr);
                    const float zz = sigmoidf_(az[i][j][e] + deg * bcz + xt[H + col] + bhz);
                    const float ghn = anh[i][j][e] + bhn;
                    const float nn = tanhf_(ani[i][j][e] + deg * bcn + xt[2 * H + col] + rr * ghn);
                    const float hp = (float)hin_hi[row * LDP + col] + (float)hin_lo[row * LDP + col];
                    s_pre[row * S::LD + col] = (1.0f - zz) * nn + zz * hp;
                    ar[i][j][e] = rr; az[i][j][e] = zz; ani[i][j][e] = nn; anh[i][j][e] = ghn;
                }
            }
        STAMP(3);
        __syncthreads();
        STAMP(4);
        // ---- C. LayerNorm row statistics
        if (has_ln) {
            for (int row = grp; row < kTileRows; row += S::GROUPS) {
                float4 v = ld4(s_pre + row * S::LD + 4 * lr);
                const float mean = group_sum<S::LPR>(v.x + v.y + v.z + v.w) * (1.0f / H);
                v = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
                const float var = group_sum<S::LPR>(dot4(v, v)) * (1.0f / H);
                const float rstd = rsqrtf(var + a.eps);
                const float4 dy = ld4(s_dy + row * S::LD + 4 * lr);
                const float4 gm = ld4(sv.lnw + 4 * lr);
                const float4 g = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
                const float c1 = group_sum<S::LPR>(g.x + g.y + g.z + g.w) * (1.0f / H);
                const float c2 = group_sum<S::LPR>(dot4(g, v)) * rstd * (1.0f / H);
                if (lr == 0) { s_stat[row * 4 + 0] = mean; s_stat[row * 4 + 1] = rstd; s_stat[row * 4 + 2] = c1; s_stat[row * 4 + 3] = c2; }
            }
            __syncthreads();
        }
        STAMP(5);
        // ---- D. LayerNorm + GRU backward in accumulator layout
        f32x4 dhd[S::RTW][S::HCW];
#pragma unroll
        for (int j = 0; j < S::HCW; ++j) {
            const int col = (wc * S::HCW + j) * 16 + r;
            const float gamma = sv.lnw[col];
            float s_lw = 0.f, s_lb = 0.f;
#pragma unroll
            for (int i = 0; i < S::RTW; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                    const float dy = s_dy[row * S::LD + col];
                    float dh;
                    if (has_ln) {
                        const float mean = s_stat[row * 4 + 0], rstd = s_stat[row * 4 + 1];
                        const float xhat = (s_pre[row * S::LD + col] - mean) * rstd;
                        s_lw += dy * xhat; s_lb += dy;
                        dh = rstd * (dy * gamma - s_stat[row * 4 + 2] - xhat * s_stat[row * 4 + 3]);
                    } else {
                        dh = dy;
                    }
                    const float rr = ar[i][j][e], zz = az[i][j][e], nn = ani[i][j][e], ghn = anh[i][j][e];
                    const float hp = (float)hin_hi[row * LDP + col] + (float)hin_lo[row * LDP + col];
                    const float dan = dh * (1.0f - zz) * (1.0f - nn * nn);
                    const float daz = dh * (hp - nn) * zz * (1.0f - zz);
                    const float dar = dan * ghn * rr * (1.0f - rr);
                    const float danr = dan * rr;
                    ar[i][j][e] = dar; az[i][j][e] = daz; ani[i][j][e] = dan; anh[i][j][e] = danr;
                    dhd[i][j][e] = dh * zz;
                }
            }
            if (has_ln) { colsum_lds_x3(s_lw, s_dlnw + col); colsum_lds_x3(s_lb, s_dlnb + col); }
        }
        STAMP(6);
        // ---- E. four gate-gradient tiles through region C (which held pre/dy until here)
        f32x4 dag[S::RTW][S::HCW];
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int j = 0; j < S::HCW; ++j) dag[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // kTPR of the four tiles (r, z, n, n*r) per round: fewer barriers, longer MFMA runs between them
#pragma unroll
        for (int rnd = 0; rnd < 4 / kTPR; ++rnd) {
            __syncthreads();          // readers of region C: phase D, or the previous round
            STAMP(14);
#pragma unroll
            for (int t2 = 0; t2 < kTPR; ++t2) {
                const int p = kTPR * rnd + t2;
                __bf16* ph = d_hi + t2 * 2 * kTileRows * LDP;      // plane set t2: {hi, lo} back to back
                __bf16* pl = ph + kTileRows * LDP;
#pragma unroll
                for (int i = 0; i < S::RTW; ++i)
#pragma unroll
                    for (int j = 0; j < S::HCW; ++j) {
                        const int col = (wc * S::HCW + j) * 16 + r;
                        const f32x4 v = p == 0 ? ar[i][j] : p == 1 ? az[i][j] : p == 2 ? ani[i][j] : anh[i][j];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            __bf16 hh, ll;
                            split_bf16(v[e], hh, ll);
                            const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                            ph[row * LDP + col] = hh;
                            pl[row * LDP + col] = ll;
                        }
                    }
            }
            STAMP(10);
            __syncthreads();
            STAMP(11);
#pragma unroll
            for (int t2 = 0; t2 < kTPR; ++t2) {
                const int p = kTPR * rnd + t2;
                const int g = p < 2 ? p : 2;
                const __bf16* ph = d_hi + t2 * 2 * kTileRows * LDP;
                const __bf16* pl = ph + kTileRows * LDP;
                if (need_dgrad) {
#pragma unroll 1
                    for (int ks = 0; ks < H / 32; ++ks) {
                        const int ko = 32 * ks + 8 * q;
                        bf16x8 xh[S::RTW], xl[S::RTW];
#pragma unroll
                        for (int i = 0; i < S::RTW; ++i) {
                            const int off = ((wr * S::RTW + i) * 16 + r) * LDP + ko;
                            xh[i] = ldfrag(ph + off); xl[i] = ldfrag(pl + off);
                        }
#pragma unroll
                        for (int j = 0; j < S::HCW; ++j) {
                            const int wo = (((wc * S::HCW + j) * 3 + g) * (H / 32) + ks) * 512 + lane * 8;   // fragment order
                            const int jc = (wc * S::HCW + j) * 16;
                            if (p != 3) {
                                const bf16x8 bh = ldfrag_tr(wc_hi + g * H * M::WLD, M::WLD, 32 * ks, jc), bl = ldfrag(a.wpack + 5 * BLK + wo);
#pragma unroll
                                for (int i = 0; i < S::RTW; ++i) mma_x3(dag[i][j], xh[i], xl[i], bh, bl);
                            }
                            if (p != 2) {
                                const bf16x8 bh = ldfrag_tr(whh_hi + g * H * M::WLD, M::WLD, 32 * ks, jc), bl = ldfrag(a.wpack + 7 * BLK + wo);
#pragma unroll
                                for (int i = 0; i < S::RTW; ++i) mma_x3(dhd[i][j], xh[i], xl[i], bh, bl);
                            }
                        }
                    }
                }
            }
            STAMP(12);
#pragma unroll
            for (int t2 = 0; t2 < kTPR; ++t2) {
                const int p = kTPR * rnd + t2;
                const int g = p < 2 ? p : 2;
                const __bf16* ph = d_hi + t2 * 2 * kTileRows * LDP;
                const __bf16* pl = ph + kTileRows * LDP;
                if constexpr (WBLK) {
                    // round 0 (r, z gates): both matrices take the tile; round 1: Wc takes the n-input tile (t2 = 0), Whh the
                    // n-hidden tile (t2 = 1), each from its own wave group in ONE pass
                    const bool whh = w >= 4;
                    if (rnd == 0) wgrad_blk_x3<H>(gWc[g], ph, pl, whh ? hin_hi : agg_hi, whh ? hin_lo : agg_lo);
                    else if (t2 == (whh ? 1 : 0)) wgrad_blk_x3<H>(gWc[2], ph, pl, whh ? hin_hi : agg_hi, whh ? hin_lo : agg_lo);
                } else {
                    if (p != 3) wgrad_x3<H>(gWc[g], ph, pl, agg_hi, agg_lo);
                    if (p != 2) wgrad_x3<H>(gWhh[g], ph, pl, hin_hi, hin_lo);
                }
                if (w < H / 16) {          // wave-uniform: gate-column tile w of the bias-type gradients
#pragma unroll
                    for (int ks = 0; ks < kTileRows / 32; ++ks)
                        mma_x3(gX[p], ldfrag_tr2(ph, LDP, 32 * ks, w * 16), ldfrag_tr2(pl, LDP, 32 * ks, w * 16),
                               ldfrag_tr2(xe_hi, XLD, 32 * ks, 0), ldfrag_tr2(xe_lo, XLD, 32 * ks, 0));
                }
            }
            STAMP(13);
        }
        STAMP(7);
        // ---- F. outputs through LDS (region C as two fp32 tiles again)
        __syncthreads();
        STAMP(8);
        if (need_dgrad) {
#pragma unroll
            for (int i = 0; i < S::RTW; ++i)
#pragma unroll
                for (int j = 0; j < S::HCW; ++j) {
                    const int col = (wc * S::HCW + j) * 16 + r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                        s_pre[row * S::LD + col] = dag[i][j][e];
                        s_dy[row * S::LD + col] = dhd[i][j][e];
                    }
                }
            __syncthreads();
            for (int row = grp; row < kTileRows; row += S::GROUPS) {
                const int64_t node = base + row;
                if (node < a.N) {
                    st4(a.g_agg_out + node * H + 4 * lr, ld4(s_pre + row * S::LD + 4 * lr));
                    st4(a.g_direct_out + node * H + 4 * lr, ld4(s_dy + row * S::LD + 4 * lr));
                }
            }
        }
        idx_commit<kThreadsX3>(idx_lds(idx_base, b ^ 1).idx, ri);
        __syncthreads();
        STAMP(9);
    }
    STAMP_FLUSH(a);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        if constexpr (WBLK) {
            wgrad_blk_flush_x3<H>(gWc[g], (w >= 4 ? a.dWhh : a.dWc) + (int64_t)g * H * H);
        } else {
            wgrad_flush_x3<H>(gWc[g], a.dWc + (int64_t)g * H * H);
            wgrad_flush_x3<H>(gWhh[g], a.dWhh + (int64_t)g * H * H);
        }
    }
    if (w < H / 16) {
        // gX[p][e] = sum_rows dG_p[row][i] * Xe[row][j] with i = 16w + 4q + e, j = r:
        //   j = 0 -> deg-weighted (dbc), j = 1..8 -> per feature class (dxtab), j = 9 -> plain column sum (dbhh)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = w * 16 + q * 4 + e;
                const float v = gX[p][e];
                const int g = p < 2 ? p : 2;
                if (p != 3) {
                    if (r == 0) atomicAdd(a.dbc + g * H + i, v);
                    if (r >= 1 && r <= 8 && r - 1 < a.C) atomicAdd(a.dxtab + (r - 1) * 3 * H + g * H + i, v);
                }
                if (p != 2 && r == 9) atomicAdd(a.dbhh + g * H + i, v);
            }
    }
    __syncthreads();
    if (has_ln)
        for (int i = tid; i < H; i += kThreadsX3) { atomicAdd(a.dlnw + i, s_dlnw[i]); atomicAdd(a.dlnb + i, s_dlnb[i]); }
}

template <int H>
int launch_fwd_x3(const StageX3Args& a, hipStream_t st) {
    const size_t shm = X3Smem<H>::fwd_bytes;
    static bool set = false;
    if (!set) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_struct_stage_fwd_x3<H>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    const int64_t ntiles = (a.N + kTileRows - 1) / kTileRows;
    int per_cu = (int)(160 * 1024 / shm);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    hipLaunchKernelGGL(k_struct_stage_fwd_x3<H>, dim3(grid_for(ntiles, per_cu)), dim3(kThreadsF), shm, st, a);
    MGV_LAUNCH_RET();
}
template <int H>
int launch_bwd_x3(const StageX3Args& a, hipStream_t st) {
    const size_t shm = X3Smem<H>::bwd_bytes;
    static bool set = false;
    if (!set) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_struct_stage_bwd_x3<H>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    const int64_t ntiles = (a.N + kTileRows - 1) / kTileRows;
    hipLaunchKernelGGL(k_struct_stage_bwd_x3<H>, dim3(grid_for(ntiles, 1)), dim3(kThreadsX3), shm, st, a);
    MGV_LAUNCH_RET();
}

}  // namespace mgv

#ifdef MGV_STAMPS
static unsigned long long* g_stamps = nullptr;
extern "C" int mgv_diag_set_stamps(void* p) { g_stamps = static_cast<unsigned long long*>(p); return 0; }
#define MGV_SET_STAMPS(a) (a).stamps = g_stamps
#else
#define MGV_SET_STAMPS(a)
#endif

static int xcd_tiles() {
    return 1;            // XCD-contiguous tile order (measured against round-robin in round 2; no switch left)
}

extern "C" int mgv_struct_stage_fwd_x3(int H, int64_t N, const float* h_in, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                                       const uint8_t* xcls, const float* xtab, int C, const void* wpack_bf16, const float* bc,
                                       const float* bhh, const float* ln_w, const float* ln_b, float ln_eps, float* h_out,
                                       int heavy_n, const int32_t* heavy_nodes, float* heavy_ws, const int32_t* table_own_idx,
                                       int nbr_tagged, float* ln_stats_out, void* stream) {
    MGV_CHECK_ARG(N >= 0 && h_in && nbr_ptr && xcls && xtab && wpack_bf16 && bc && bhh && h_out && (table_own_idx == nullptr || !nbr_tagged || N < (1 << 24)));
    MGV_CHECK_ARG(heavy_n >= 0 && (heavy_n == 0 || (heavy_nodes && heavy_ws)));
    MGV_CHECK_ARG(C >= 1 && C <= mgv::kMaxClsX3);
    MGV_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr));
    if (N == 0) return MGV_OK;
    MGV_CHECK_ARG(nbr_idx != nullptr);
    mgv::StageX3Args a{};
    a.N = N; a.h_in = h_in; a.ptr = nbr_ptr; a.idx = nbr_idx; a.xcls = xcls; a.xtab = xtab; a.C = C;
    a.wpack = static_cast<const __bf16*>(wpack_bf16); a.bc = bc; a.bhh = bhh; a.lnw = ln_w; a.lnb = ln_b; a.eps = ln_eps; a.h_out = h_out;
    a.gmask = -1;
    if (table_own_idx) { a.own_idx = table_own_idx; if (nbr_tagged) { a.hshift = 24; a.gmask = 0xffffff; } }
    a.ln_stats = ln_w ? ln_stats_out : nullptr;
    MGV_SET_STAMPS(a);
    a.xcd = xcd_tiles();
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (H) {
        case 32: mgv::launch_heavy_sums<32>(a, heavy_n, heavy_nodes, heavy_ws, false, st); return mgv::launch_fwd_x3<32>(a, st);
        case 64: mgv::launch_heavy_sums<64>(a, heavy_n, heavy_nodes, heavy_ws, false, st); return mgv::launch_fwd_x3<64>(a, st);
        default: return MGV_EUNSUPPORTED;
    }
}

extern "C" int mgv_struct_stage_bwd_x3(int H, int64_t N, const float* h_in, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                                       const uint8_t* xcls, const float* xtab, int C, const void* wpack_bf16, const float* bc,
                                       const float* bhh, const float* ln_w, const float* ln_b, float ln_eps,
                                       const float* gy_direct, const float* gy_agg, float* g_direct_out, float* g_agg_out,
                                       float* dWc, float* dbc, float* dWhh, float* dbhh, float* dxtab, float* dln_w,
                                       float* dln_b, int heavy_n, const int32_t* heavy_nodes, float* heavy_ws,
                                       const int32_t* table_own_idx, int nbr_tagged, void* stream) {
    MGV_CHECK_ARG(N >= 0 && h_in && nbr_ptr && xcls && xtab && wpack_bf16 && bc && bhh && gy_direct && (table_own_idx == nullptr || !nbr_tagged || N < (1 << 24)));
    MGV_CHECK_ARG(heavy_n >= 0 && (heavy_n == 0 || (heavy_nodes && heavy_ws)));
    MGV_CHECK_ARG(dWc && dbc && dWhh && dbhh && dxtab);
    MGV_CHECK_ARG(C >= 1 && C <= mgv::kMaxClsX3);
    MGV_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr));
    MGV_CHECK_ARG(ln_w == nullptr || (dln_w && dln_b));
    MGV_CHECK_ARG((g_direct_out == nullptr) == (g_agg_out == nullptr));
    if (N == 0) return MGV_OK;
    MGV_CHECK_ARG(nbr_idx != nullptr);
    mgv::StageX3Args a{};
    a.N = N; a.h_in = h_in; a.ptr = nbr_ptr; a.idx = nbr_idx; a.xcls = xcls; a.xtab = xtab; a.C = C;
    a.wpack = static_cast<const __bf16*>(wpack_bf16); a.bc = bc; a.bhh = bhh; a.lnw = ln_w; a.lnb = ln_b; a.eps = ln_eps;
    a.gy_direct = gy_direct; a.gy_agg = gy_agg; a.g_direct_out = g_direct_out; a.g_agg_out = g_agg_out;
    a.dWc = dWc; a.dbc = dbc; a.dWhh = dWhh; a.dbhh = dbhh; a.dxtab = dxtab; a.dlnw = dln_w; a.dlnb = dln_b;
    a.gmask = -1;
    if (table_own_idx) { a.own_idx = table_own_idx; if (nbr_tagged) { a.hshift = 24; a.gmask = 0xffffff; } }
    MGV_SET_STAMPS(a);
    a.xcd = xcd_tiles();
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (H) {
        case 32: mgv::launch_heavy_sums<32>(a, heavy_n, heavy_nodes, heavy_ws, gy_agg != nullptr, st); return mgv::launch_bwd_x3<32>(a, st);
        case 64: mgv::launch_heavy_sums<64>(a, heavy_n, heavy_nodes, heavy_ws, gy_agg != nullptr, st); return mgv::launch_bwd_x3<64>(a, st);
        default: return MGV_EUNSUPPORTED;
    }
}

#if MGV_ABLF != 0
// marker of a timing-ablation build (wrong results by design): deepgate/_hip.py refuses a library that exports it
extern "C" int mgv_diag_ablation_build_fwd(void) { return MGV_ABLF; }
#endif
