// Backward of a structural-encoder half round (digae_layer.py:266-275 under autograd), second decomposition, H = 64.
//
// The first backward kernel (struct_stage_x3.hip) streamed half of every weight fragment from L2 inside its MFMA loops
// and ran its phases at ~2.5x their MFMA time.  Here NO weight byte moves after the prologue:
//
//   * wave (wc, m) of the 8-wave workgroup owns hidden-column tile wc (16 columns x 3 gates) of ONE matrix
//     (m = 0: Wc against the neighbour sums, m = 1: Whh against the own rows) for all 64 rows of a tile.  Its
//     recompute fragments (12) and its dgrad fragments (12) are 96 VGPRs and stay in registers for the whole kernel.
//   * every product is computed TRANSPOSED (weights as the MFMA A operand, node rows as B): a lane then holds ONE node
//     and four CONSECUTIVE hidden columns, so biases / class rows / dY / own rows are 16-byte LDS reads, the
//     gate-gradient planes are written with 8-byte stores, LayerNorm statistics need two cross-lane steps and a
//     64 x 4 x 16 B exchange between the four column waves, and the outputs leave as 16-byte LDS stores.
//   * the two matrices' pre-activations meet through a lane-linear LDS exchange (each wave hands the partner the two
//     row tiles whose epilogue the partner runs): 48 KB per tile, conflict-free 16-byte accesses.
//   * all four gate-gradient planes (r, z, n, n*r; hi + lo) are resident at once: one dgrad + wgrad phase per tile,
//     five workgroup barriers per tile (was twelve), one of them behind the row gather; outputs leave straight from the accumulators.
//   * parameter gradients leave through per-workgroup slabs and a fixed-order reduction kernel: no float atomics,
//     bit-identical from run to run.
#include <cstddef>
#include <type_traits>
#include "struct_stage_x3_common.h"
#ifndef MGV_BWD2_D
#define MGV_BWD2_D 3            // neighbour slots per row and gather round (2 x 3 x 2 row loads in flight per lane; 2: 1-2 % slower, same box)
#endif
#ifndef MGV_WG1
#define MGV_WG1 0           // experiment (round 4): weight gradients on ONE bf16 product per term (hi x hi) instead of the three of bf16x3:
                            // -6.7 % per launch, but 2.5e-3 of the gradient scale (a sum of N noise-like terms keeps the 2^-8 relative error
                            // of its products: it does not average out) against the 1e-3 gradient bar: not usable; builds with it are refused
#endif
#ifndef MGV_ABL
#define MGV_ABL 0            // timing ablations of diagnostic builds (results are wrong): 1 no MFMA, 2 light VALU in P2/P3, 4 no row gathers, 8 no output stores, 16 dgrad weights loaded once, 32 (with 1) no LDS fragment reads
#endif

namespace mgv {

#if (MGV_ABL & 33) == 33
#define mma_x3(c, ah, al, bh, bl) ((void)0)                  // and no LDS fragment reads either
#elif MGV_ABL & 1
#define mma_x3(c, ah, al, bh, bl) asm volatile("" :: "v"(ah), "v"(al), "v"(bh), "v"(bl))
#endif


struct B2 {
    static constexpr int H = 64;
    static constexpr int LDP = H + 8;                         // bf16 elements per plane row
    static constexpr int PB = kTileRows * LDP * 2;            // bytes of one plane
    static constexpr int LDF = H + 4;                         // floats per fp32 tile row
    static constexpr int F32TILE = kTileRows * LDF * 4;
    static constexpr int SMALL_F = kMaxClsX3 * 3 * H + 3 * H + 3 * H + H + H + kTileRows + kTileRows;
    static constexpr int o_x = 0;                             // agg_hi, agg_lo, hin_hi, hin_lo
    static constexpr int o_c = o_x + 4 * PB;                  // region C: exchange (48 KB) -> dG planes (72 KB) -> output tiles (34 KB)
    static constexpr int C_BYTES = 8 * PB;
    static constexpr int EX_BYTES = kNW * 6 * 1024;
    static_assert(EX_BYTES <= C_BYTES && 2 * F32TILE <= C_BYTES, "region C");
    static constexpr int o_dy = o_c + C_BYTES;                // fp32 dY tile, own region: the next tile's row phase may start while outputs drain
    static constexpr int o_dhz = o_dy + F32TILE;              // dh*z of row tiles 0,1 from the m = 0 waves to the m = 1 waves (lane-linear)
    static constexpr int o_small = o_dhz + 4 * 2 * 1024;
    static constexpr int o_part = o_small + SMALL_F * 4;      // LayerNorm partials [64 rows][4 column waves] float4
    static constexpr int o_idx = o_part + kTileRows * 4 * 16;
    static constexpr int IDX_BYTES = 2 * kIdxStride * 4;
    static constexpr int o_xe = o_idx + IDX_BYTES;            // xe_hi, xe_lo [64][XLD]
    static constexpr int o_lnacc = o_xe + 2 * kTileRows * XLD * 2;    // per-wave LayerNorm affine gradient sums [8 waves][2][16 columns]
    static constexpr int o_stat = o_lnacc + kNW * 2 * 16 * 4;     // {mean, rstd} of the tile's rows when the forward kept them (StageX3Args::ln_stats)
    static constexpr int bytes = o_stat + kTileRows * 2 * 4;
    static_assert(bytes <= 160 * 1024, "LDS budget");
    // per-workgroup gradient slab (floats): [8 waves][14 float4 slots][64 lanes] then dlnw[64], dlnb[64]
    static constexpr int SLOTS = 14;                          // 12 weight-gradient tiles, 2 bias-type tiles
    static constexpr int SLAB_W = kNW * SLOTS * 64 * 4;
    static constexpr int SLAB = SLAB_W + 2 * H;
};

__device__ __forceinline__ float quad_rows_sum(float v) {      // sum over the four lanes r, r+16, r+32, r+48
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

struct B2Args {
    StageX3Args s;
    float* slab;        // [gridDim][B2::SLAB]
};

// The kernel's arguments, re-read from the kernarg segment (scalar loads) behind an opaque copy of its address.  Used at the head of
// every phase of the tile loop: the ~20 pointers and sizes the loop touches then live in scalar registers for one phase each instead
// of for the whole kernel, where half of them were spilled to vector-register lanes and came back through v_readlane in the vector
// pipe (54 spilled SGPRs, ~110 v_readlane / v_writelane per tile and wave).
// kargs_now() assumes the by-value B2Args is the kernel's ONLY parameter (kernarg offset 0, host layout) with StageX3Args first
static_assert(offsetof(B2Args, s) == 0 && std::is_trivially_copyable<B2Args>::value && alignof(B2Args) <= 16, "kargs_now(): B2Args layout");
__device__ __forceinline__ const StageX3Args& kargs_now() {
    const B2Args __attribute__((address_space(4)))* p = (const B2Args __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return ((const B2Args*)p)->s;
}

__global__ __launch_bounds__(kThreadsX3) void k_struct_stage_bwd2_x3(B2Args args);
static_assert(std::is_same<decltype(&k_struct_stage_bwd2_x3), void (*)(B2Args)>::value, "kargs_now(): single by-value B2Args parameter");
__global__ __launch_bounds__(kThreadsX3) void k_struct_stage_bwd2_x3(B2Args args) {
    const StageX3Args& a = args.s;
    constexpr int H = B2::H, LDP = B2::LDP, LDF = B2::LDF, BLK = 3 * H * H, PE = kTileRows * B2::LDP;   // PE: elements of one plane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* agg_hi = reinterpret_cast<__bf16*>(smem_raw + B2::o_x);
    __bf16* agg_lo = agg_hi + PE;
    __bf16* hin_hi = agg_hi + 2 * PE;
    __bf16* hin_lo = agg_hi + 3 * PE;
    f32x4* s_ex = reinterpret_cast<f32x4*>(smem_raw + B2::o_c);
    __bf16* s_dg = reinterpret_cast<__bf16*>(smem_raw + B2::o_c);           // plane p (r, z, n, n*r): hi at 2p planes, lo one plane later
    float* s_out_agg = reinterpret_cast<float*>(smem_raw + B2::o_c);
    float* s_out_dir = reinterpret_cast<float*>(smem_raw + B2::o_c + B2::F32TILE);
    float* s_dy = reinterpret_cast<float*>(smem_raw + B2::o_dy);
    f32x4* s_dhz = reinterpret_cast<f32x4*>(smem_raw + B2::o_dhz);
    const SmallVecs sv = stage_small<H>(a, reinterpret_cast<float*>(smem_raw + B2::o_small));
    fold_bhh_rz<H>(a, sv);
    f32x4* s_part = reinterpret_cast<f32x4*>(smem_raw + B2::o_part);
    int* idx_base = reinterpret_cast<int*>(smem_raw + B2::o_idx);
    __bf16* xe_hi = reinterpret_cast<__bf16*>(smem_raw + B2::o_xe);
    __bf16* xe_lo = xe_hi + kTileRows * XLD;
    float* s_lnacc = reinterpret_cast<float*>(smem_raw + B2::o_lnacc);
    float* s_stat = reinterpret_cast<float*>(smem_raw + B2::o_stat);

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wc = w & 3, m = w >> 2;     // wave-uniform: scalar registers
    // Lane-derived indices are re-derived per phase behind an opaque copy of the thread id (LANE_IDS): otherwise every LDS
    // address of the loop body is hoisted out of the tile loop and the ~40 loop-invariant address registers are spilled.
#define LANE_IDS \
    int t_ = tid; asm volatile("" : "+v"(t_)); \
    const int lane = t_ & 63, r = lane & 15, q = lane >> 4, grp = t_ >> 4, lr = t_ & 15, c0 = 16 * wc + 4 * q; \
    (void)lane; (void)r; (void)q; (void)grp; (void)lr; (void)c0;
    const bool has_ln = a.lnw != nullptr;
    const bool has_stats = has_ln && a.ln_stats != nullptr;
    const bool need_dgrad = a.g_direct_out != nullptr;
    const float ln_eps = a.eps;
    const int64_t ntiles = (a.N + kTileRows - 1) / kTileRows;

    // Weight fragments (fragment-order pack: block, (row tile, k-step) of 512 elements, lane l at 8 l) are re-read from L2
    // every tile in two bursts whose latency is covered: the 12 recompute fragments ([k-step][gate], rows = gate columns of
    // tile wc, k = input features) in front of the row gathers, the 12 dgrad fragments in front of the wgrad MFMAs.
    const int lane0 = tid & 63;
    const __bf16* wr_hi_p_ = a.wpack + (2 * m) * BLK + wc * 2 * 512 + lane0 * 8;
    const __bf16* wr_lo_p_ = a.wpack + (2 * m + 1) * BLK + wc * 2 * 512 + lane0 * 8;
    // dgrad fragments (rows = output columns of tile wc, k = the 192 gate columns): 12 KB per wave, re-read from L2 every
    // tile in ONE burst issued in front of the weight-gradient MFMAs that cover its latency
    // (scalar base + 32-bit lane offset: global_load with an SGPR address, no 64-bit pointer registers to keep alive)
    const int wd_off = (4 + 2 * m) * BLK + wc * 6 * 512;

    const __bf16 *wr_hi_p = wr_hi_p_, *wr_lo_p = wr_lo_p_;
    // ---- persistent accumulators
    f32x4 gW[3][4];                         // 2x2 block of this wave's matrix, per gate (wgrad_blk_x3)
    f32x4 gX[2];                            // bias-type gradients of planes p = m (pp 0) and 2 + m (pp 1), gate-column tile wc
    // LayerNorm affine gradients: summed over the 16 nodes of a lane group in registers, then added by one lane to this wave's
    // own LDS slots (program order: deterministic); 8 VGPRs less to keep alive across the whole kernel
    for (int i = tid; i < kNW * 2 * 16; i += kThreadsX3) s_lnacc[i] = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) gW[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    gX[0] = gX[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (static priority for the second-dispatched half, waves 4-7, only swaps which half waits at the barriers: measured zero-sum)
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
    // recompute fragments of the first tile; every later tile's set is requested while the previous tile's outputs drain
    bf16x8 wr_hi[2][3], wr_lo[2][3];
    {
        asm volatile("" : "+v"(wr_hi_p), "+v"(wr_lo_p));      // opaque per tile: keeps the (loop-invariant) loads inside the loop
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                wr_hi[ks][g] = ldfrag_global(wr_hi_p + (g * 8 + ks) * 512);
                wr_lo[ks][g] = ldfrag_global(wr_lo_p + (g * 8 + ks) * 512);
            }
    }
    STAMP_DECL
    for (int it = 0; seq.at(it) < ntiles; ++it, b ^= 1) {
        const int64_t base = seq.at(it) * kTileRows;
        STAMP_BEGIN;
        // ---- P0. row phase: gather, sum, split into the operand planes
        {
            const StageX3Args& a = kargs_now();
            LANE_IDS
            float4 acc[2], own[2], dy[2];
            float deg[2];
            int cls[2];
            float st_r[2] = {0.f, 0.f};          // the rows' kept LayerNorm statistics: requested with the row loads, parked behind barrier (0)
            if (has_stats && lr < 2) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int64_t node = base + grp + rr * 32;
                    if (node < a.N) st_r[rr] = a.ln_stats[node * 2 + lr];
                }
            }
#if MGV_ABL & 4
            for (int rr = 0; rr < 2; ++rr) { acc[rr] = make_float4(0.1f * lr, 0.2f, 0.3f, 0.4f); own[rr] = acc[rr]; dy[rr] = acc[rr]; deg[rr] = 2.f; cls[rr] = 1; }
#else
            tile_rows<H, 2, true, MGV_BWD2_D>(a, base, grp, 32, lr, idx_lds(idx_base, b).ptr, idx_lds(idx_base, b).idx, *idx_lds(idx_base, b).dmax(), acc, own, dy, deg, cls);
#endif
            // (0) the previous tile's P4 (last reader of the planes and of xe) is over in every wave; placed here, behind the
            //     gather, it waits where the waves wait for memory anyway
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = grp + rr * 32;
                bf16x4 hi, lo;
                split4(acc[rr], hi, lo);
                st_bf4(agg_hi + row * LDP + 4 * lr, hi); st_bf4(agg_lo + row * LDP + 4 * lr, lo);
                split4(own[rr], hi, lo);
                st_bf4(hin_hi + row * LDP + 4 * lr, hi); st_bf4(hin_lo + row * LDP + 4 * lr, lo);
                st4(s_dy + row * LDF + 4 * lr, dy[rr]);
                if (lr == 0) { sv.deg[row] = deg[rr]; sv.cls[row] = cls[rr]; }
                if (has_stats && lr < 2) s_stat[row * 2 + lr] = st_r[rr];
                const float xe = lr == 0 ? deg[rr] : (lr <= 8 ? (cls[rr] == lr - 1 ? 1.0f : 0.0f) : (lr == 9 ? 1.0f : 0.0f));
                __bf16 xh, xl;
                split_bf16(xe, xh, xl);
                xe_hi[row * XLD + lr] = xh; xe_lo[row * XLD + lr] = xl;
            }
        }
        if (tid <= kTileRows) idx_lds(idx_base, b ^ 1).ptr[tid] = rp;
        STAMP(0);
        __syncthreads();                                    // (1) planes, dY, deg/cls, next tile's pointers
        STAMP(1);
        // next tile's indices: requested now, parked in LDS before barrier (6)
        {
            const StageX3Args& a = kargs_now();
            idx_prefetch<kThreadsX3>(a, idx_lds(idx_base, b ^ 1).ptr, ri);
            rp = ptr_prefetch(a, seq.at(it + 2), ntiles);
        }
        tile_dmax(idx_lds(idx_base, b ^ 1).ptr, idx_lds(idx_base, b ^ 1).dmax());
        // ---- P1. pre-activations of this wave's matrix, transposed: lane (r, q) <- node 16 i + r, columns c0..c0+3
        f32x4 oa[3][2];                                     // own row tiles 2m, 2m+1: [gate][il]
        {
            LANE_IDS
            const __bf16* xh = m ? hin_hi : agg_hi;
            const __bf16* xl = m ? hin_lo : agg_lo;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int i0 = half == 0 ? 2 * (1 - m) : 2 * m;       // the partner's two row tiles first: they leave through LDS
#pragma unroll
                for (int g = 0; g < 3; ++g) { oa[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; oa[g][1] = oa[g][0]; }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int il = 0; il < 2; ++il) {
                        const int off = ((i0 + il) * 16 + r) * LDP + 32 * ks + 8 * q;
                        const bf16x8 fh = ldfrag(xh + off), fl = ldfrag(xl + off);
#pragma unroll
                        for (int g = 0; g < 3; ++g) mma_x3(oa[g][il], wr_hi[ks][g], wr_lo[ks][g], fh, fl);
                    }
                if (half == 0) {
                    const int wp = wc + 4 * (1 - m);
#pragma unroll
                    for (int il = 0; il < 2; ++il)
#pragma unroll
                        for (int g = 0; g < 3; ++g) s_ex[(wp * 6 + il * 3 + g) * 64 + lane] = oa[g][il];
                }
            }
        }
        idx_commit<kThreadsX3>(idx_lds(idx_base, b ^ 1).idx, ri);          // next tile's indices (requested when P1 began)
        STAMP(2);
        __syncthreads();                                    // (2) exchange, next tile's indices
        STAMP(3);
        // ---- L2 prefetch of the NEXT tile's rows: one 4-byte load per 128-byte line (own + dY rows, then the neighbour rows of both
        //      gathered arrays), at most two per lane, at the head of the two LDS/VALU-only phases: no other global load is issued
        //      for the next ~7,000 cycles, so nothing queues behind them on the in-order vmcnt.  The next row phase then gathers from
        //      the XCD's L2 instead of the fabric.  The loaded words only feed a comparison that never holds.
        unsigned pf0 = 0, pf1 = 0;
        if (seq.at(it + 1) < ntiles) {
          const StageX3Args& a = kargs_now();
          if (a.prefetch) {
            const int64_t nb = seq.at(it + 1) * kTileRows;
            const int* n_ptr = idx_lds(idx_base, b ^ 1).ptr;
            const int* n_idx = idx_lds(idx_base, b ^ 1).idx;
            const int ne2 = 2 * min(n_ptr[kTileRows] - n_ptr[0], 128);      // neighbour lines (2 per row); longer lists: the head only
            const bool two = a.gy_agg != nullptr;
            // wave-uniform roles (scalar base pointers, 32-bit lane offsets): waves 0-1 own rows, waves 2-3 dY rows, waves 4-7
            // neighbour rows of h_in; second slot: neighbour rows of gy_agg on waves 0-3
            // buffer loads: scalar resource + 32-bit byte offset per lane (no 64-bit address registers to keep alive)
            const unsigned nbytes = (unsigned)min((int64_t)a.N * H * 4, (int64_t)0xfffffff0);
            if (w < 4) {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w < 2 ? a.h_in : a.gy_direct), 0, nbytes, 0x00020000);
                const int t = tid & 127;
                int64_t row = nb + (t >> 1);
                row = row < a.N ? row : a.N - 1;
                if (w >= 2 || a.hshift == 0)        // (table mode: h_in is the small class table, nothing to prefetch and N rows are not behind it)
                    pf0 = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(row * (H * 4)) + 128u * (t & 1), 0, 0);     // default cache policy: the line must stay in L2
                if (two && tid < ne2) {
                    const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy_agg), 0, nbytes, 0x00020000);
                    pf1 = __builtin_amdgcn_raw_buffer_load_b32(rg, (unsigned)(n_idx[tid >> 1] & a.gmask) * (H * 4) + 128u * (tid & 1), 0, 0);
                }
            } else {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.h_in), 0, nbytes, 0x00020000);
                const int t = tid - 256;
                if (t < ne2 && a.hshift == 0) pf0 = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)n_idx[t >> 1] * (H * 4) + 128u * (t & 1), 0, 0);
            }
          }
        }
        // ---- P2. GRU forward values of row tiles 2m, 2m+1 and LayerNorm partials over this wave's 16 columns.  Written on 4-vectors
        //      (two packed fp32 operations each); the r and z gates only need Wc.agg + Whh.h, so the partner's half and this wave's
        //      half are added without asking which is which (only the n gate must tell them apart); b_hr, b_hz sit in the class table
        f32x4 vr[2], vz[2], vn[2], vg[2], vd[2];            // r, z, n, Whh_n h + b_hn, pre - (wave mean)
        float mw[2];
#pragma unroll
        for (int il = 0; il < 2; ++il) {
            LANE_IDS
            const int row = 16 * (2 * m + il) + r;
            const f32x4 rx0 = s_ex[(w * 6 + il * 3 + 0) * 64 + lane], rx1 = s_ex[(w * 6 + il * 3 + 1) * 64 + lane], rx2 = s_ex[(w * 6 + il * 3 + 2) * 64 + lane];
            const f32x4 sr = rx0 + oa[0][il], sz = rx1 + oa[1][il];
            const f32x4 pn = m ? rx2 : oa[2][il], hn = m ? oa[2][il] : rx2;
#if MGV_ABL & 2
            vr[il] = rx0 + oa[0][il]; vz[il] = rx1 + oa[1][il]; vn[il] = rx2 + oa[2][il]; vg[il] = vr[il]; vd[il] = vz[il]; mw[il] = (float)row;
            continue;
#endif
            const float deg = sv.deg[row];
            const float* xt = sv.xtab + sv.cls[row] * 3 * H + c0;
            const bf16x4 hh = *reinterpret_cast<const bf16x4*>(hin_hi + row * LDP + c0);
            const bf16x4 hl = *reinterpret_cast<const bf16x4*>(hin_lo + row * LDP + c0);
            const f32x4 bcr = ldv4(sv.bc + c0), bcz = ldv4(sv.bc + H + c0), bcn = ldv4(sv.bc + 2 * H + c0);
            const f32x4 xr = ldv4(xt), xz = ldv4(xt + H), xn = ldv4(xt + 2 * H), bhn = ldv4(sv.bhh + 2 * H + c0);
            const f32x4 rr = sigmoid4(sr + (deg * bcr + xr));
            const f32x4 zz = sigmoid4(sz + (deg * bcz + xz));
            const f32x4 ghn = hn + bhn;
            const f32x4 nn = tanh4(pn + (deg * bcn + xn) + rr * ghn);
            const f32x4 hp = f32x4{(float)hh[0], (float)hh[1], (float)hh[2], (float)hh[3]} + f32x4{(float)hl[0], (float)hl[1], (float)hl[2], (float)hl[3]};
            const f32x4 pre = nn + zz * (hp - nn);
            vr[il] = rr; vz[il] = zz; vn[il] = nn; vg[il] = ghn;
            const float s1 = (pre[0] + pre[1]) + (pre[2] + pre[3]);
            if (has_stats) {
                // mean and rstd come from the forward: the row sums left are the two that depend on dY
                const float mean = s_stat[row * 2], rstd = s_stat[row * 2 + 1];
                const f32x4 dy = ldv4(s_dy + row * LDF + c0), gm = ldv4(sv.lnw + c0);
                const f32x4 xh = (pre - mean) * rstd, gg = dy * gm;
                const f32x4 gd = gg * xh;
                vd[il] = xh;
                const float s3 = quad_rows_sum((gg[0] + gg[1]) + (gg[2] + gg[3]));
                const float s4 = quad_rows_sum((gd[0] + gd[1]) + (gd[2] + gd[3]));
                mw[il] = rstd;
                if (q == 0) s_part[row * 4 + wc] = f32x4{s3, s4, 0.f, 0.f};
            } else if (has_ln) {
                const float mean_w = quad_rows_sum(s1) * (1.0f / 16.0f);
                const f32x4 dy = ldv4(s_dy + row * LDF + c0), gm = ldv4(sv.lnw + c0);
                const f32x4 d = pre - mean_w, gg = dy * gm;
                const f32x4 dd = d * d, gd = gg * d;
                vd[il] = d;
                const float m2 = quad_rows_sum((dd[0] + dd[1]) + (dd[2] + dd[3]));
                const float s3 = quad_rows_sum((gg[0] + gg[1]) + (gg[2] + gg[3]));
                const float s4 = quad_rows_sum((gd[0] + gd[1]) + (gd[2] + gd[3]));
                mw[il] = mean_w;
                if (q == 0) s_part[row * 4 + wc] = f32x4{mean_w, m2, s3, s4};
            } else {
                vd[il] = pre; mw[il] = 0.f;
            }
        }
        STAMP(4);
        __syncthreads();                                    // (3) LayerNorm partials; every exchange slot has been read
        STAMP(5);
        // ---- P3. LayerNorm + GRU backward; gate gradients to the planes
        f32x4 dhz[2];
        float lnw_acc[4] = {0.f, 0.f, 0.f, 0.f}, lnb_acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int il = 0; il < 2; ++il) {
            LANE_IDS
            const int row = 16 * (2 * m + il) + r;
#if MGV_ABL & 2
            {
                __bf16* dst = s_dg + row * LDP + c0;
                const bf16x4 h0 = bf16x4{(__bf16)vr[il][0], (__bf16)vr[il][1], (__bf16)vr[il][2], (__bf16)vr[il][3]};
                const bf16x4 h1 = bf16x4{(__bf16)vz[il][0], (__bf16)vz[il][1], (__bf16)vz[il][2], (__bf16)vz[il][3]};
                const bf16x4 h2 = bf16x4{(__bf16)vn[il][0], (__bf16)vn[il][1], (__bf16)vn[il][2], (__bf16)vn[il][3]};
                st_bf4(dst, h0); st_bf4(dst + PE, h1); st_bf4(dst + 2 * PE, h2); st_bf4(dst + 3 * PE, h0);
                st_bf4(dst + 4 * PE, h1); st_bf4(dst + 5 * PE, h2); st_bf4(dst + 6 * PE, h0); st_bf4(dst + 7 * PE, h1);
                dhz[il] = vd[il] + mw[il];
                if (m == 0) s_dhz[(wc * 2 + il) * 64 + lane] = dhz[il];
                continue;
            }
#endif
            const float4 dy = ld4(s_dy + row * LDF + c0);
            const float dy_[4] = {dy.x, dy.y, dy.z, dy.w};
            const bf16x4 hh = *reinterpret_cast<const bf16x4*>(hin_hi + row * LDP + c0);
            const bf16x4 hl = *reinterpret_cast<const bf16x4*>(hin_lo + row * LDP + c0);
            float dh[4];
            if (has_stats) {
                const f32x4 p0 = s_part[row * 4 + 0], p1 = s_part[row * 4 + 1], p2 = s_part[row * 4 + 2], p3 = s_part[row * 4 + 3];
                const float rstd = mw[il];
                const float c1 = (p0[0] + p1[0] + p2[0] + p3[0]) * (1.0f / H);
                const float c2 = (p0[1] + p1[1] + p2[1] + p3[1]) * (1.0f / H);
                const float4 gm = ld4(sv.lnw + c0);
                const float gm_[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xhat = vd[il][e];
                    lnw_acc[e] += dy_[e] * xhat; lnb_acc[e] += dy_[e];
                    dh[e] = rstd * (dy_[e] * gm_[e] - c1 - xhat * c2);
                }
            } else if (has_ln) {
                const f32x4 p0 = s_part[row * 4 + 0], p1 = s_part[row * 4 + 1], p2 = s_part[row * 4 + 2], p3 = s_part[row * 4 + 3];
                const float mean = (p0[0] + p1[0] + p2[0] + p3[0]) * 0.25f;
                const float e0 = p0[0] - mean, e1 = p1[0] - mean, e2 = p2[0] - mean, e3 = p3[0] - mean;
                const float var = (p0[1] + p1[1] + p2[1] + p3[1] + 16.0f * (e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3)) * (1.0f / H);
                const float rstd = rsqrtf(var + ln_eps);
                const float c1 = (p0[2] + p1[2] + p2[2] + p3[2]) * (1.0f / H);
                const float c2 = (p0[3] + p1[3] + p2[3] + p3[3] + e0 * p0[2] + e1 * p1[2] + e2 * p2[2] + e3 * p3[2]) * rstd * (1.0f / H);
                const float4 gm = ld4(sv.lnw + c0);
                const float gm_[4] = {gm.x, gm.y, gm.z, gm.w};
                const float shift = mw[il] - mean;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xhat = (vd[il][e] + shift) * rstd;
                    lnw_acc[e] += dy_[e] * xhat; lnb_acc[e] += dy_[e];      // this lane's two nodes first, ONE cross-lane sum per tile
                    dh[e] = rstd * (dy_[e] * gm_[e] - c1 - xhat * c2);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) dh[e] = dy_[e];
            }
            float dar[4], daz[4], dan[4], danr[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float rr = vr[il][e], zz = vz[il][e], nn = vn[il][e], ghn = vg[il][e];
                const float hp = (float)hh[e] + (float)hl[e];
                dan[e] = dh[e] * (1.0f - zz) * (1.0f - nn * nn);
                daz[e] = dh[e] * (hp - nn) * zz * (1.0f - zz);
                dar[e] = dan[e] * ghn * rr * (1.0f - rr);
                danr[e] = dan[e] * rr;
                dhz[il][e] = dh[e] * zz;
            }
            bf16x4 hi, lo;
            __bf16* dst = s_dg + row * LDP + c0;
            split4(make_float4(dar[0], dar[1], dar[2], dar[3]), hi, lo);
            st_bf4(dst, hi); st_bf4(dst + PE, lo);
            split4(make_float4(daz[0], daz[1], daz[2], daz[3]), hi, lo);
            st_bf4(dst + 2 * PE, hi); st_bf4(dst + 3 * PE, lo);
            split4(make_float4(dan[0], dan[1], dan[2], dan[3]), hi, lo);
            st_bf4(dst + 4 * PE, hi); st_bf4(dst + 5 * PE, lo);
            split4(make_float4(danr[0], danr[1], danr[2], danr[3]), hi, lo);
            st_bf4(dst + 6 * PE, hi); st_bf4(dst + 7 * PE, lo);
            if (m == 0) s_dhz[(wc * 2 + il) * 64 + lane] = dhz[il];
        }
        if (has_ln) {
            LANE_IDS
            float lw_[4], lb_[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { lw_[e] = group_sum<16>(lnw_acc[e]); lb_[e] = group_sum<16>(lnb_acc[e]); }
            if (r == 0) {
                float* acc = s_lnacc + w * 32 + 4 * q;
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e] += lw_[e]; acc[16 + e] += lb_[e]; }
            }
        }
        if ((pf0 ^ pf1) == 0x7fc12345u && a.stamps) a.stamps[0] = pf0;     // keeps the prefetch loads alive; never true in practice
        STAMP(6);
        __syncthreads();                                    // (4) gate-gradient planes, dh*z hand-off, next tile's indices
        STAMP(7);
        // ---- P4. weight gradients and dgrad, interleaved in two halves: each half's six dgrad fragments leave L2 in front of
        //      weight-gradient MFMAs that cover the latency (12 KB per wave and tile; 24 VGPRs at a time)
        f32x4 dgo[4];
        {
            const StageX3Args& a = kargs_now();
            LANE_IDS
            const __bf16* x_hi = m ? hin_hi : agg_hi;
            const __bf16* x_lo = m ? hin_lo : agg_lo;
            // Weight gradients of all three gates in ONE pass over the tile's two 32-row k-steps: the transposed fragments of the
            // wave's two input-column tiles (and of the [deg, onehot, 1] columns) are read once per k-step and serve every gate;
            // the bias-type tile (gX) reuses the gate-gradient fragment the 2x2 block loads anyway (its row tile wc is it0 + (wc & 1),
            // its planes p = 2 pp + m are among the three this wave's matrix reads).  The dgrad fragments of each half leave L2 in
            // front of one k-step of these MFMAs.
            const int it0 = 2 * (wc >> 1), jt0 = 2 * (wc & 1), ig = wc & 1;
            bf16x8 wd_hi[3], wd_lo[3];
            int oz = 0;
#if !(MGV_ABL & 16)
            asm volatile("" : "+s"(oz));                    // opaque per tile: keeps the (loop-invariant) loads inside the loop
#endif
            const __bf16* wd_p = a.wpack + wd_off + oz;
            if (need_dgrad) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { wd_hi[k] = ldfrag_global(wd_p + k * 512 + lane * 8); wd_lo[k] = ldfrag_global(wd_p + BLK + k * 512 + lane * 8); }
            }
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                const int k0 = 32 * half;
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) { bh[j] = ldfrag_tr2(x_hi, LDP, k0, (jt0 + j) * 16); if (!MGV_WG1) bl[j] = ldfrag_tr2(x_lo, LDP, k0, (jt0 + j) * 16); }
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int p = g == 2 ? 2 + m : g;
                    const __bf16* ph = s_dg + p * 2 * PE;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const bf16x8 ah = ldfrag_tr2(ph, LDP, k0, (it0 + i) * 16), al = ldfrag_tr2(ph + PE, LDP, k0, (it0 + i) * 16);
#pragma unroll
                        for (int j = 0; j < 2; ++j) { if (MGV_WG1) gW[g][i * 2 + j] = mfma_bf16(ah, bh[j], gW[g][i * 2 + j]); else mma_x3(gW[g][i * 2 + j], ah, al, bh[j], bl[j]); }
                        if (i == ig && (g == 2 || g == m)) mma_x3(gX[g == 2 ? 1 : 0], ah, al, ldfrag_tr2(xe_hi, XLD, k0, 0), ldfrag_tr2(xe_lo, XLD, k0, 0));
                    }
                }
            }
            STAMP(8);
            if (need_dgrad) {
                // seeds of the accumulators (after the weight gradients: 16 registers less to carry through them)
                if (m == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dgo[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    dgo[0] = s_dhz[(wc * 2 + 0) * 64 + lane]; dgo[1] = s_dhz[(wc * 2 + 1) * 64 + lane];
                    dgo[2] = dhz[0]; dgo[3] = dhz[1];
                }
                // first half of the gate columns for all four row tiles
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const __bf16* ph = s_dg + (k >> 1) * 2 * PE;
                    const __bf16* pl = ph + PE;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int off = (i * 16 + r) * LDP + 32 * (k & 1) + 8 * q;
                        mma_x3(dgo[i], wd_hi[k], wd_lo[k], ldfrag(ph + off), ldfrag(pl + off));
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) { wd_hi[k] = ldfrag_global(wd_p + (3 + k) * 512 + lane * 8); wd_lo[k] = ldfrag_global(wd_p + BLK + (3 + k) * 512 + lane * 8); }
                // second half row tile by row tile; each finished accumulator leaves at once, straight from the registers: lane (r, q)
                // holds 16 contiguous bytes of row 16 i + r, the four column waves complete each 256-byte row within the same phase
                // and L2 merges the 64-byte pieces.  No staging tile, no barrier behind it: the next row phase writes planes / dY /
                // xe / deg,cls only behind its barrier (0), region C is rewritten after its barrier (1).
                float* go = m ? a.g_direct_out : a.g_agg_out;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int ks = 3 + k;
                        const int p = (ks >> 1) == 2 ? 2 + m : (ks >> 1);
                        const __bf16* ph = s_dg + p * 2 * PE;
                        const int off = (i * 16 + r) * LDP + 32 * (ks & 1) + 8 * q;
                        mma_x3(dgo[i], wd_hi[k], wd_lo[k], ldfrag(ph + off), ldfrag(ph + PE + off));
                    }
                    const int64_t node = base + i * 16 + r;
#if MGV_ABL & 8
                    if (node < a.N && dgo[i][0] == 1.2345e-30f) *reinterpret_cast<f32x4*>(go + node * H + c0) = dgo[i];
#else
                    if (node < a.N) *reinterpret_cast<f32x4*>(go + node * H + c0) = dgo[i];
#endif
                }
            }
        }
        STAMP(9);
        STAMP(13);
        // no barrier: the next row phase writes planes / dY / xe / deg,cls (all dead since (5)); region C is rewritten
        // only after the next tile's barrier (1)
    }
    STAMP_FLUSH(a);

    // ---- flush: per-workgroup slab, lane-linear float4 slots (k_struct_stage_bwd2_reduce knows the mapping)
    LANE_IDS
    float* slab = args.slab + (int64_t)blockIdx.x * B2::SLAB;
    f32x4* sw = reinterpret_cast<f32x4*>(slab) + (w * B2::SLOTS) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) sw[(g * 4 + t) * 64] = gW[g][t];
    sw[12 * 64] = gX[0];
    sw[13 * 64] = gX[1];
    // LayerNorm affine gradients: the two row halves (m) in a fixed order
    __syncthreads();
    if (tid < 2 * H) {
        const int which = tid >> 6, c = tid & 63, wcc = c >> 4, cc = c & 15;
        slab[B2::SLAB_W + which * H + c] = s_lnacc[wcc * 32 + which * 16 + cc] + s_lnacc[(4 + wcc) * 32 + which * 16 + cc];
    }
}

// Fixed-order sum of the workgroup slabs into the gradient accumulators (+=): 64 slots per block, 4 slab phases per slot,
// LDS tree over the phases.  Deterministic; 30 MB read once.
struct B2RedArgs {
    const float* slab; int nwg; int C;
    float* dWc; float* dbc; float* dWhh; float* dbhh; float* dxtab; float* dlnw; float* dlnb;
};

__global__ __launch_bounds__(256) void k_struct_stage_bwd2_reduce(B2RedArgs a) {
    constexpr int H = B2::H;
    __shared__ f32x4 red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tx;                       // float4 slot index: [wave][slot][lane], then the 32 LayerNorm float4s
    constexpr int NW4 = B2::SLAB_W / 4, NT = NW4 + 2 * H / 4;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < NT) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.slab) + t;
#pragma unroll 8
        for (int g = ty; g < a.nwg; g += 4) {
            const f32x4 v = src[(int64_t)g * (B2::SLAB / 4)];
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty != 0 || t >= NT) return;
    f32x4 v = red[0][tx];
#pragma unroll
    for (int k = 1; k < 4; ++k) { v[0] += red[k][tx][0]; v[1] += red[k][tx][1]; v[2] += red[k][tx][2]; v[3] += red[k][tx][3]; }
    if (t >= NW4) {                                           // dlnw[64] then dlnb[64]
        const int c = (t - NW4) * 4;
        float* dst = c < H ? a.dlnw + c : a.dlnb + (c - H);
        if (a.dlnw)
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] += v[e];
        return;
    }
    const int lane = t & 63, slot = (t >> 6) % B2::SLOTS, w = t / (64 * B2::SLOTS);
    const int r = lane & 15, q = lane >> 4, wc = w & 3, m = w >> 2;
    if (slot < 12) {                                          // gW[g][i*2+j]: rows = gate columns (it0+i)*16 + 4q + e, column (jt0+j)*16 + r
        const int g = slot >> 2, tt = slot & 3, i = tt >> 1, j = tt & 1;
        const int it0 = 2 * (wc >> 1), jt0 = 2 * (wc & 1);
        float* dW = (m ? a.dWhh : a.dWc) + (int64_t)g * H * H;
#pragma unroll
        for (int e = 0; e < 4; ++e) dW[((it0 + i) * 16 + q * 4 + e) * H + (jt0 + j) * 16 + r] += v[e];
    } else {                                                  // gX of plane p: gate column wc*16 + 4q + e against xe column r
        const int p = 2 * (slot - 12) + m, g = p < 2 ? p : 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = wc * 16 + q * 4 + e;
            if (p != 3) {
                if (r == 0) a.dbc[g * H + i] += v[e];
                if (r >= 1 && r <= 8 && r - 1 < a.C) a.dxtab[(r - 1) * 3 * H + g * H + i] += v[e];
            }
            if (p != 2 && r == 9) a.dbhh[g * H + i] += v[e];
        }
    }
}

// fixed-order sum of `grid` workgroup slabs into the gradient accumulators (shared with struct_stage_bwd3_x3.hip: same slab layout)
int launch_stage_slab_reduce(const StageX3Args& s, float* workspace, int grid, hipStream_t st) {
    B2RedArgs ra{workspace, grid, s.C, s.dWc, s.dbc, s.dWhh, s.dbhh, s.dxtab, s.dlnw, s.dlnb};
    constexpr int NT = B2::SLAB_W / 4 + 2 * B2::H / 4;
    hipLaunchKernelGGL(k_struct_stage_bwd2_reduce, dim3((NT + 63) / 64), dim3(256), 0, st, ra);
    MGV_LAUNCH_RET();
}

int launch_bwd2_x3(const StageX3Args& s, float* workspace, int64_t workspace_floats, hipStream_t st) {
    static bool set[64] = {false};      // per device: each device loads its own copy of the code object
    int dev = 0;
    hipGetDevice(&dev);
    if (!set[dev & 63]) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_struct_stage_bwd2_x3), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set[dev & 63] = true; }
    const int64_t ntiles = (s.N + kTileRows - 1) / kTileRows;
    const int grid = grid_for(ntiles, 1);
    if (workspace == nullptr || workspace_floats < (int64_t)grid * B2::SLAB) return MGV_EINVAL;
    B2Args a{s, workspace};
    hipLaunchKernelGGL(k_struct_stage_bwd2_x3, dim3(grid), dim3(kThreadsX3), B2::bytes, st, a);
    return launch_stage_slab_reduce(s, workspace, grid, st);
}

}  // namespace mgv

#ifdef MGV_STAMPS
static unsigned long long* g_stamps2 = nullptr;
extern "C" int mgv_diag_set_stamps2(void* p) { g_stamps2 = static_cast<unsigned long long*>(p); return 0; }
#define MGV_SET_STAMPS2(a) (a).stamps = g_stamps2
#else
#define MGV_SET_STAMPS2(a)
#endif

extern "C" int mgv_struct_stage_bwd2_ws_floats(int H, int64_t N) {
    if (H != 64 || N < 0) return 0;
    const int64_t ntiles = (N + mgv::kTileRows - 1) / mgv::kTileRows;
    return mgv::grid_for(ntiles, 1) * mgv::B2::SLAB;      // <= 256 workgroups x 28,800 floats
}

extern "C" int mgv_struct_stage_bwd2_x3(int H, int64_t N, const float* h_in, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                                        const uint8_t* xcls, const float* xtab, int C, const void* wpack_bf16, const float* bc,
                                        const float* bhh, const float* ln_w, const float* ln_b, float ln_eps,
                                        const float* gy_direct, const float* gy_agg, float* g_direct_out, float* g_agg_out,
                                        float* dWc, float* dbc, float* dWhh, float* dbhh, float* dxtab, float* dln_w,
                                        float* dln_b, float* workspace, int64_t workspace_floats, int heavy_n,
                                        const int32_t* heavy_nodes, float* heavy_ws, const int32_t* table_own_idx,
                                        int nbr_tagged, const float* ln_stats, void* stream) {
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
    MGV_CHECK_ARG(table_own_idx == nullptr || !nbr_tagged || N < (1 << 24));
    if (table_own_idx) { a.own_idx = table_own_idx; if (nbr_tagged) { a.hshift = 24; a.gmask = 0xffffff; } }
    a.ln_stats = ln_w ? const_cast<float*>(ln_stats) : nullptr;
    MGV_SET_STAMPS2(a);
    a.xcd = 1;           // XCD-contiguous tile order and the L2 row prefetch: both measured, no switch left
    a.prefetch = (table_own_idx && !nbr_tagged) ? 0 : 1;      // (own rows through an index into a shorter h_in: the row prefetch assumes N rows behind h_in)
    MGV_CHECK_ARG(heavy_n >= 0 && (heavy_n == 0 || (heavy_nodes && heavy_ws)));
    mgv::launch_heavy_sums<64>(a, heavy_n, heavy_nodes, heavy_ws, gy_agg != nullptr, static_cast<hipStream_t>(stream));
    return mgv::launch_bwd2_x3(a, workspace, workspace_floats, static_cast<hipStream_t>(stream));
}

#if MGV_ABL != 0 || MGV_WG1 != 0
// marker of a timing-ablation build (wrong results by design): deepgate/_hip.py refuses a library that exports it
extern "C" int mgv_diag_ablation_build(void) { return MGV_ABL; }
#endif
