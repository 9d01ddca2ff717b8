// Levelised functional sweep on bf16x3 split-precision MFMA (mgv_x3.h): same operator and phases as
// func_level.hip, whose header describes the math.  Differences:
//   * the zbar tile and the gate gradients live in LDS as bf16 hi/lo planes; recompute, dgrad and wgrad run on
//     v_mfma_f32_16x16x32_bf16; the wgrad reads both operands transposed (ds_read_b64_tr_b16) from the same
//     row-major planes; weights arrive pre-split in MFMA fragment order, per slot
//         Wvc_hi | Wvc_lo   ([3H][2H], blocks (row tile, k-step))   forward / recompute B operands
//         WvcT_hi | WvcT_lo ([2H][3H])                               dgrad B operands
//   * a level is ONE round of tiles on the chip (<= 2 per CU at the baseline shapes), so a level costs one
//     tile's dependent-load chain.  The chain is cut to  span -> edge lists -> rows : a tile first stages its
//     rows' CSR spans (order_span, one 16-byte load per row), then its in-/out-edge lists and the per-edge
//     scalars (alpha, d score, consumer gate slot) in LDS with one thread per edge, and only then gathers
//     rows, all of a row's loads in flight together.  8 waves per workgroup, two rows per 16-lane group.
#include "func_level_x3_common.h"

#ifndef MGV_LVL_FWD_KU
#define MGV_LVL_FWD_KU 4        // k-steps whose weight fragments the forward requests together (1: four dependent L2 trips per tile; 4: one, 100 VGPRs)
#endif

namespace mgv {

template <int H, bool HID = false>
__global__ __launch_bounds__(kLT, 4) void k_level_fwd_x3(LevelX3Args a) {
    using S = SplitL<H>;
    using M = LvlSmem<H>;
    constexpr int LPR = M::LPR, GROUPS = M::GROUPS, RPG = M::RPG, LDO = M::LDO;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* z_hi = reinterpret_cast<__bf16*>(smem_raw + M::o_zhi);
    __bf16* z_lo = reinterpret_cast<__bf16*>(smem_raw + M::o_zlo);
    float* s_o = reinterpret_cast<float*>(smem_raw + M::o_zhi);       // output tile, after the MFMAs
    const LvlIdx ix = lvl_idx(smem_raw + M::o_idx_f);
    const int tile = a.tile_begin + blockIdx.x;
    STAMP_DECL
    STAMP_BEGIN;
    const int start = a.tile_start[tile], count = a.tile_count[tile], g = a.tile_slot[tile];
    stage_spans(a, start, count, ix);
    const LvlSmall sv = lvl_small<H>(a, g, reinterpret_cast<float*>(smem_raw + M::o_small_f));
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
    const int grp = tid / LPR, lr = tid % LPR;
    lds_barrier();
    STAMP(0);
    if (a.span_ints != kRowInts) {       // (packed rows brought the in-edge sources with the spans)
        stage_in_edges(a, ix);
        lds_barrier();
    }
    STAMP(1);
    const float4 us = ld4(sv.u + 4 * lr), uf = ld4(sv.u + H + 4 * lr);
    {
        InRows<H> L[RPG];
        int4 sp[RPG];
#pragma unroll
        for (int i = 0; i < RPG; ++i) {
            const int row = grp + i * GROUPS;
            sp[i] = ix.span[row];
            L[i].issue(a, ix.insrc + row * kInCap, sp[i].y - sp[i].x, lr);
        }
#pragma unroll
        for (int i = 0; i < RPG; ++i) {
            const int row = grp + i * GROUPS;
            float m, inv;
            float4 zs, zf;
            attn_reduce<H>(a, L[i], sp[i], us, uf, lr, m, inv, zs, zf);
            store_zbar<H>(z_hi, z_lo, row, lr, zs, zf);
            if (lr == 0) sv.sa[row] = sp[i].y > sp[i].x ? 1.0f : 0.0f;
        }
    }
    STAMP(2);
    lds_barrier();
    STAMP(3);
    f32x4 ar[S::RTW], az[S::RTW], an[S::RTW];
    lvl_gemm_x3<H, MGV_LVL_FWD_KU>(a.wpack + (int64_t)g * 4 * 6 * H * H, z_hi, z_lo, ar, az, an);
    STAMP(4);
    lds_barrier();                   // s_o overlays the planes
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
                if constexpr (HID) {
                    // previous-round state: gh = W_hh h + b_hh per gate block, h' = (1 - z) n + z h
                    const int64_t node = ix.node[row];
                    float gr = 0.f, gz = 0.f, gn = 0.f, hp = 0.f;
                    if (node >= 0) { const float* g_ = a.gh + node * 3 * H + col; gr = g_[0]; gz = g_[H]; gn = g_[2 * H]; hp = a.hprev[node * H + col]; }
                    const float rr = sigmoidf_(ar[i][e] + sa * bvr + cr + gr);
                    const float zz = sigmoidf_(az[i][e] + sa * bvz + cz + gz);
                    const float nn = tanhf_(an[i][e] + sa * bvn + cn + rr * (gn + bhn));
                    s_o[row * LDO + col] = (1.0f - zz) * nn + zz * hp;
                } else {
                    const float rr = sigmoidf_(ar[i][e] + sa * bvr + cr);
                    const float zz = sigmoidf_(az[i][e] + sa * bvz + cz);
                    const float nn = tanhf_(an[i][e] + sa * bvn + cn + rr * bhn);
                    s_o[row * LDO + col] = (1.0f - zz) * nn;        // h0 = 0
                }
            }
    }
    STAMP(5);
    lds_barrier();
#pragma unroll
    for (int i = 0; i < RPG; ++i) {
        const int row = grp + i * GROUPS;
        if (row < count) st4(a.hf + (int64_t)ix.node[row] * H + 4 * lr, ld4(s_o + row * LDO + 4 * lr));
    }
    STAMP(6);
    STAMP_FLUSH(a);
}


// Backward of one tile.  Global memory discipline: a wave's loads wait for every older load, store or atomic of
// that wave (vmcnt is in order), so nothing is stored before the tile's last load has been issued; per-tile float
// atomics are out (memory-side, ~1.3 TB/s chip-wide, 14x slower when every workgroup adds to the same rows).  The
// weight gradient is therefore not formed here: the tile leaves its gate gradients dG[pos][3H] and its zbar rows
// z[pos][2H] (pos = position in `order`) for k_sweep_wgrad_x3, and adds its small parameter gradients (gu, dbvc,
// dbih, dbhh) with plain loads/stores into the slab that workgroup b of EVERY level owns per slot.
template <int H, bool HID = false>
__global__ __launch_bounds__(kLT, 4) void k_level_bwd_x3(LevelX3Args a) {
    using S = SplitL<H>;
    using S2 = SplitL<2 * H>;
    using M = LvlSmem<H>;
    constexpr int LPR = M::LPR, GROUPS = M::GROUPS, RPG = M::RPG, LDO = M::LDO;
    constexpr int LDZP = M::LDZP, LDGP = M::LDGP, LDZF = M::LDZF, BLK = 6 * H * H;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* z_hi = reinterpret_cast<__bf16*>(smem_raw + M::o_zhi);
    __bf16* z_lo = reinterpret_cast<__bf16*>(smem_raw + M::o_zlo);
    float* s_dh = reinterpret_cast<float*>(smem_raw + M::o_r);
    __bf16* d_hi = reinterpret_cast<__bf16*>(smem_raw + M::o_r);
    __bf16* d_lo = d_hi + kTileRows * LDGP;
    float* s_dz = reinterpret_cast<float*>(smem_raw + M::o_r);
    const OutStage os = out_stage(smem_raw + M::o_out);
    const LvlIdx ix = lvl_idx(smem_raw + M::o_idx_b);
    const int tile = a.tile_begin + blockIdx.x;
    STAMP_DECL
    STAMP_BEGIN;
    const int start = a.tile_start[tile], count = a.tile_count[tile], g = a.tile_slot[tile];
    stage_spans(a, start, count, ix);
    const LvlSmall sv = lvl_small<H>(a, g, reinterpret_cast<float*>(smem_raw + M::o_small_b));
    float* s_gu = reinterpret_cast<float*>(smem_raw + M::o_acc);
    float* s_dbvc = s_gu + 2 * H; float* s_dbih = s_dbvc + 3 * H; float* s_dbhh = s_dbih + 3 * H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
    const int grp = tid / LPR, lr = tid % LPR;
    // small parameter gradients of this tile start from ZERO in LDS: each bias column then receives exactly two addends (the two
    // row-tile waves of its column tile; a + b == b + a bit for bit), and the workgroup's slab row is read-modified-written at
    // the end — same result whichever wave adds first
    float* slab_small = a.wslab + ((int64_t)blockIdx.x * a.T + g) * (11 * H);
    for (int i = tid; i < 11 * H; i += kLT) s_gu[i] = 0.f;
    const bool packed = a.span_ints == kRowInts;
    const int out_cap = packed ? kRowOut : kOutCap;          // consumers per row staged in LDS (the rest: pull_tail)
    if (packed) {
        stage_out_edges_packed(a, start, count, os);         // spans, in-edge sources, consumers and their scalars: one staging phase
    } else {
        lds_barrier();
        STAMP(0);
        stage_in_edges(a, ix);
        stage_out_edges(a, ix, os);
    }
    lds_barrier();
    STAMP(1);
    const float4 us = ld4(sv.u + 4 * lr), uf = ld4(sv.u + H + 4 * lr);
    // ---- 0/1. pull dL/dhf and dL/dhs of the tile's nodes from their consumers, recompute their attention
#ifndef MGV_LVL_EARLY_Z
#define MGV_LVL_EARLY_Z 1
#endif
#ifndef MGV_LVL_EARLY_GHS
#define MGV_LVL_EARLY_GHS 1     // dL/dhs rows leave in the pull phase (8 registers less across the MFMA phases: 22 -> 10 spilled, backward sweep 7.24 -> 6.78 ms)
#endif
#if !MGV_LVL_EARLY_GHS
    float4 gs_keep[2] = {zero4(), zero4()};      // dL/dhs rows, stored at the end
#endif
    static_assert(RPG <= 2, "two kept rows");
#pragma unroll 1
    for (int i = 0; i < RPG; ++i) {
        const int row = grp + i * GROUPS;
        const int4 sp = ix.span[row];
        const int node = ix.node[row];
        // a gate with a very long consumer list took its pull in this level's pre-pass (mgv_func_sweep_bwd_x3): look it up
        int hk = -1;
        if (a.heavy_k1 > a.heavy_k0 && sp.w - sp.z > a.skip_active) {
            int lo = a.heavy_k0, hi = a.heavy_k1 - 1;
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1, v = a.heavy_nodes[mid];
                if (v == node) { hk = mid; break; }
                if (v < node) lo = mid + 1; else hi = mid - 1;
            }
        }
        const int nout = hk >= 0 ? 0 : min(sp.w - sp.z, out_cap);
        InRows<H> L;
        OutRows<H> P;
        f32x4 own;
        L.issue(a, ix.insrc + row * kInCap, sp.y - sp.x, lr);
        P.issue(a, os, row * kOutCap, 0, nout, lr);
        if (node >= 0) own = *reinterpret_cast<const f32x4*>(a.ghf + (int64_t)node * H + 4 * lr);
        float4 gs = zero4(), gf = zero4();
        P.reduce(os, sv.uall, row * kOutCap, 0, nout, lr, gs, gf);
        for (int k0 = kOutChunk; k0 < nout; k0 += kOutChunk) {
            OutRows<H> Q;
            Q.issue(a, os, row * kOutCap, k0, nout, lr);
            Q.reduce(os, sv.uall, row * kOutCap, k0, nout, lr, gs, gf);
        }
        if (hk >= 0) {
            gs = ld4(a.heavy_pull + (int64_t)hk * 2 * H + 4 * lr);
            gf = ld4(a.heavy_pull + (int64_t)hk * 2 * H + H + 4 * lr);
        } else {
            pull_tail<H>(a, sp.z + out_cap, sp.w, lr, gs, gf);
        }
        float4 dh = zero4();
        if (node >= 0) dh = add4(gf, f4(own));
#if MGV_LVL_EARLY_GHS
        if (row < count) st4(a.ghs + (int64_t)node * H + 4 * lr, gs);
#else
        if (i == 0) gs_keep[0] = gs; else gs_keep[1] = gs;
#endif
        float m, inv;
        float4 zs, zf;
        attn_reduce<H>(a, L, sp, us, uf, lr, m, inv, zs, zf);
        store_zbar<H>(z_hi, z_lo, row, lr, zs, zf);
#if MGV_LVL_EARLY_Z
        if (row < count) {               // the zbar row for the deferred weight gradient leaves here, in fp32 as formed (not re-read from the planes)
            st4(a.zrows + (int64_t)(start + row) * 2 * H + 4 * lr, zs);
            st4(a.zrows + (int64_t)(start + row) * 2 * H + H + 4 * lr, zf);
        }
#endif
        st4(s_dh + row * LDO + 4 * lr, dh);
        if (lr == 0) { sv.sa[row] = sp.y > sp.x ? 1.0f : 0.0f; sv.m[row] = m; sv.inv[row] = inv; }
    }
    STAMP(2);
    lds_barrier();
    STAMP(3);
    // ---- 2. recompute gates
    const __bf16* wslot = a.wpack + (int64_t)g * 4 * BLK;
    f32x4 ar[S::RTW], az[S::RTW], an[S::RTW];
    lvl_gemm_x3<H>(wslot, z_hi, z_lo, ar, az, an);
    STAMP(4);
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
                float dar, daz, dan, rr;
                if constexpr (HID) {
                    const int64_t node = ix.node[row];
                    float gr = 0.f, gz = 0.f, gn = 0.f, hp = 0.f;
                    if (node >= 0) { const float* g_ = a.gh + node * 3 * H + col; gr = g_[0]; gz = g_[H]; gn = g_[2 * H]; hp = a.hprev[node * H + col]; }
                    rr = sigmoidf_(ar[i][e] + sa * bvr + cr + gr);
                    const float zz = sigmoidf_(az[i][e] + sa * bvz + cz + gz);
                    const float ghn = gn + bhn;
                    const float nn = tanhf_(an[i][e] + sa * bvn + cn + rr * ghn);
                    const float dh = s_dh[row * LDO + col];
                    dan = dh * (1.0f - zz) * (1.0f - nn * nn);
                    daz = dh * (hp - nn) * zz * (1.0f - zz);
                    dar = dan * ghn * rr * (1.0f - rr);
                    if (node >= 0) {
                        float* d_ = a.dgh + node * 3 * H + col;
                        d_[0] = dar; d_[H] = daz; d_[2 * H] = dan * rr;       // d(gh): the caller's linear kernels turn it into d(W_hh), d(b_hh), d(h_prev)
                        a.ghprev[node * H + col] = dh * zz;                   // the direct path to the previous state
                    }
                } else {
                    rr = sigmoidf_(ar[i][e] + sa * bvr + cr);
                    const float zz = sigmoidf_(az[i][e] + sa * bvz + cz);
                    const float nn = tanhf_(an[i][e] + sa * bvn + cn + rr * bhn);
                    const float dh = s_dh[row * LDO + col];
                    dan = dh * (1.0f - zz) * (1.0f - nn * nn);
                    daz = -dh * nn * zz * (1.0f - zz);
                    dar = dan * bhn * rr * (1.0f - rr);
                }
                ar[i][e] = dar; az[i][e] = daz; an[i][e] = dan;
                b_r += dar; b_z += daz; b_n += dan; h_n += dan * rr;
                v_r += sa * dar; v_z += sa * daz; v_n += sa * dan;
            }
        colsum_lds_lx(b_r, s_dbih + col); colsum_lds_lx(b_z, s_dbih + H + col); colsum_lds_lx(b_n, s_dbih + 2 * H + col);
        colsum_lds_lx(b_r, s_dbhh + col); colsum_lds_lx(b_z, s_dbhh + H + col); colsum_lds_lx(h_n, s_dbhh + 2 * H + col);
        colsum_lds_lx(v_r, s_dbvc + col); colsum_lds_lx(v_z, s_dbvc + H + col); colsum_lds_lx(v_n, s_dbvc + 2 * H + col);
    }
    STAMP(5);
    // ---- 4. three passes: d(zbar) += dG_p * Wvc[p]
    const int wc2 = w % S2::WPC, wr2 = w / S2::WPC;
    f32x4 dz[S2::RTW];
#pragma unroll
    for (int i = 0; i < S2::RTW; ++i) dz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        // B operands of this pass: issued before the barriers, their latency hides behind the plane writes
        bf16x8 bh[H / 32], bl[H / 32];
#pragma unroll
        for (int ks = 0; ks < H / 32; ++ks) {
            const int wo = ((wc2 * 3 + p) * (H / 32) + ks) * 512 + lane * 8;
            bh[ks] = ldfrag(wslot + 2 * BLK + wo); bl[ks] = ldfrag(wslot + 3 * BLK + wo);
        }
        lds_barrier();                 // readers of region R: phase 3 (dh), or the previous pass
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
        lds_barrier();
        STAMP(6);
#pragma unroll
        for (int ks = 0; ks < H / 32; ++ks)
#pragma unroll
            for (int i = 0; i < S2::RTW; ++i) {
                const int off = ((wr2 * S2::RTW + i) * 16 + r) * LDGP + 32 * ks + 8 * q;
                mma_x3(dz[i], ldfrag(d_hi + off), ldfrag(d_lo + off), bh[ks], bl[ks]);
            }
        STAMP(7);
    }
#ifndef MGV_LVL_EARLY_DG
#define MGV_LVL_EARLY_DG 1        // with the early dL/dhs store and the re-read attention vector: 22 -> 2 spilled registers, backward sweep 7.4 -> 6.85 ms
#endif
#if MGV_LVL_EARLY_DG
    // (behind the passes, not at the end of the tile: the 24 gate-gradient registers are free during the attention backward)
    // gate gradients of the tile's rows, for the weight-gradient kernel
    {
        const int col = wc * 16 + r;
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                if (row < count) {
                    float* dg = a.dgrows + (int64_t)(start + row) * 3 * H + col;
                    dg[0] = ar[i][e]; dg[H] = az[i][e]; dg[2 * H] = an[i][e];
                }
            }
    }
#endif
    // ---- 5. d(zbar) tile to LDS (fp32, row layout for the attention backward); it overlays the dG planes
    lds_barrier();
#pragma unroll
    for (int i = 0; i < S2::RTW; ++i) {
        const int col = wc2 * 16 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) s_dz[((wr2 * S2::RTW + i) * 16 + q * 4 + e) * LDZF + col] = dz[i][e];
    }
    lds_barrier();
    STAMP(9);
    // ---- 6. attention backward per in-edge: leave alpha, d(score) for the sources' pulls, d(zbar) per node.
    // A row's stores are issued after the next row's loads.
    float4 gus = zero4(), guf = zero4();
    {
        // (the attention vector re-read from LDS: 8 registers not carried across the MFMA phases)
        int lr6 = lr;
        asm volatile("" : "+v"(lr6));
        const float4 us6 = ld4(sv.u + 4 * lr6), uf6 = ld4(sv.u + H + 4 * lr6);
        InRows<H> L[RPG];
        int4 sp[RPG];
        float al[RPG][kInRegs], ds[RPG][kInRegs];
        sp[0] = ix.span[grp];
        L[0].issue(a, ix.insrc + grp * kInCap, sp[0].y - sp[0].x, lr);
#pragma unroll
        for (int i = 0; i < RPG; ++i) {
            const int row = grp + i * GROUPS;
            const float4 dzs = ld4(s_dz + row * LDZF + 4 * lr), dzf = ld4(s_dz + row * LDZF + H + 4 * lr);
#if !MGV_LVL_EARLY_Z
            const bf16x4 zsh = *reinterpret_cast<const bf16x4*>(z_hi + row * LDZP + 4 * lr), zsl = *reinterpret_cast<const bf16x4*>(z_lo + row * LDZP + 4 * lr);
            const bf16x4 zfh = *reinterpret_cast<const bf16x4*>(z_hi + row * LDZP + H + 4 * lr), zfl = *reinterpret_cast<const bf16x4*>(z_lo + row * LDZP + H + 4 * lr);
            const float4 zs = make_float4((float)zsh[0] + (float)zsl[0], (float)zsh[1] + (float)zsl[1], (float)zsh[2] + (float)zsl[2], (float)zsh[3] + (float)zsl[3]);
            const float4 zf = make_float4((float)zfh[0] + (float)zfl[0], (float)zfh[1] + (float)zfl[1], (float)zfh[2] + (float)zfl[2], (float)zfh[3] + (float)zfl[3]);
#endif
            if (row < count) attn_bwd_row<H>(a, L[i], sp[i], us6, uf6, dzs, dzf, sv.m[row], sv.inv[row], lr, al[i], ds[i], gus, guf);
            if (i == 0) { STAMP(12); } else { STAMP(14); }
            if (i + 1 < RPG) {
                sp[i + 1] = ix.span[row + GROUPS];
                L[i + 1].issue(a, ix.insrc + (row + GROUPS) * kInCap, sp[i + 1].y - sp[i + 1].x, lr);
            }
            if (row < count) {
                const int64_t node = ix.node[row];
                const int64_t pos = start + row;
                st4(a.dzb + node * 2 * H + 4 * lr, dzs);
                st4(a.dzb + node * 2 * H + H + 4 * lr, dzf);
#if !MGV_LVL_EARLY_Z
                st4(a.zrows + pos * 2 * H + 4 * lr, zs);
                st4(a.zrows + pos * 2 * H + H + 4 * lr, zf);
#endif
#if !MGV_LVL_EARLY_GHS
                st4(a.ghs + node * H + 4 * lr, gs_keep[i]);
#endif
                const int deg = sp[i].y - sp[i].x;
                if (lr == 0) {
#pragma unroll
                    for (int k = 0; k < kInRegs; ++k)
                        if (k < deg) { a.alpha[sp[i].x + k] = al[i][k]; a.dsc[sp[i].x + k] = ds[i][k]; }
                }
            }
            if (i == 0) { STAMP(13); }
        }
    }
    STAMP(10);
#if !MGV_LVL_EARLY_DG
    // gate gradients of the tile's rows, for the weight-gradient kernel
    {
        const int col = wc * 16 + r;
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                if (row < count) {
                    float* dg = a.dgrows + (int64_t)(start + row) * 3 * H + col;
                    dg[0] = ar[i][e]; dg[H] = az[i][e]; dg[2 * H] = an[i][e];
                }
            }
    }
#endif
    // d(attention vector): the lane groups of a wave that share a column quad meet by shuffles (fixed tree), the eight waves through
    // an LDS stage over the (now dead) zbar planes, summed in wave order — no LDS float atomics, bit-reproducible
    {
        constexpr int GPW = 64 / LPR;                     // lane groups per wave
#pragma unroll
        for (int mk = LPR; mk < 64; mk <<= 1) {
            gus.x += __shfl_xor(gus.x, mk, 64); gus.y += __shfl_xor(gus.y, mk, 64); gus.z += __shfl_xor(gus.z, mk, 64); gus.w += __shfl_xor(gus.w, mk, 64);
            guf.x += __shfl_xor(guf.x, mk, 64); guf.y += __shfl_xor(guf.y, mk, 64); guf.z += __shfl_xor(guf.z, mk, 64); guf.w += __shfl_xor(guf.w, mk, 64);
        }
        (void)GPW;
        lds_barrier();                                   // every wave is done with the zbar planes
        float* s_stage = reinterpret_cast<float*>(smem_raw + M::o_zhi);      // [kLW][2H]
        if (lane < LPR) {
            st4(s_stage + w * 2 * H + 4 * lane, gus);
            st4(s_stage + w * 2 * H + H + 4 * lane, guf);
        }
        lds_barrier();
        for (int i = tid; i < 11 * H; i += kLT) {
            float v = s_gu[i];
            if (i < 2 * H) {
#pragma unroll
                for (int ww = 0; ww < kLW; ++ww) v += s_stage[ww * 2 * H + i];
            }
            slab_small[i] += v;
        }
    }
    STAMP(11);
    STAMP_FLUSH(a);
}

// One segment of a heavy never-updated node's consumer list per workgroup (a primary input that drives thousands of gates): the
// hs-half of the pull, sum over consumers c of alpha_e dzb[c][:H] + dsc_e u_g(c)[:H]; 16 lane groups strided, LDS sum in group order.
template <int H, bool BOTH>
__global__ __launch_bounds__(256) void k_pull_heavy_seg(int s0, int S, const int32_t* seg_e0, const int32_t* seg_e1, const int32_t* out_dst,
                                                        const int32_t* out_slot, const uint8_t* gslot, const float* alpha, const float* dsc,
                                                        const float* dzb, const float* attn_u, float* partial) {
    constexpr int LPR = H / 4, G = 256 / LPR, W = BOTH ? 2 * H : H;      // BOTH: the hs half and the hf half (an updated gate needs both)
    __shared__ __attribute__((aligned(16))) float s_p[G][W];
    const int lr = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    for (int sg = s0 + blockIdx.x; sg < s0 + S; sg += gridDim.x) {
        float4 gs = zero4(), gf = zero4();
        for (int e = seg_e0[sg] + grp; e < seg_e1[sg]; e += G) {
            const int64_t c = out_dst[e];
            const int gc = gslot[c];
            if (gc == kNoGateX) continue;
            const int sl = out_slot[e];
            const float al = alpha[sl], ds = dsc[sl];
            gs = fma4(al, ld4(dzb + c * 2 * H + 4 * lr), fma4(ds, ld4(attn_u + (int64_t)gc * 2 * H + 4 * lr), gs));
            if (BOTH) gf = fma4(al, ld4(dzb + c * 2 * H + H + 4 * lr), fma4(ds, ld4(attn_u + (int64_t)gc * 2 * H + H + 4 * lr), gf));
        }
        __syncthreads();
        st4(&s_p[grp][4 * lr], gs);
        if (BOTH) st4(&s_p[grp][H + 4 * lr], gf);
        __syncthreads();
        if (grp == 0) {
            float4 tot = zero4(), tof = zero4();
            for (int g = 0; g < G; ++g) { tot = add4(tot, ld4(&s_p[g][4 * lr])); if (BOTH) tof = add4(tof, ld4(&s_p[g][H + 4 * lr])); }
            st4(partial + (int64_t)sg * W + 4 * lr, tot);
            if (BOTH) st4(partial + (int64_t)sg * W + H + 4 * lr, tof);
        }
    }
}

// hf rows of the nodes the sweep never updates stay zero (dg_ae_model_aig.py:61: hf starts as zeros): written here, one
// sixteenth of the rows at the baseline shapes, instead of zero-filling the whole [N, H] state in front of the sweep
__global__ __launch_bounds__(kThreads) void k_zero_inactive_rows(int64_t N, int H, const uint8_t* gslot, float* hf) {
    const int lpr = H / 4, lr = threadIdx.x % lpr;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / lpr);
    for (int64_t node = (int64_t)blockIdx.x * (kThreads / lpr) + threadIdx.x / lpr; node < N; node += stride)
        if (gslot[node] == kNoGateX) st4(hf + node * H + 4 * lr, zero4());
}

// dWvc[g] += sum over the rows of slot g of dG[row]^T zbar[row], from the rows the level kernels left behind.
// Persistent workgroups walk the slot's tile list; a tile's fp32 rows are split into bf16 hi/lo planes in LDS
// and both MFMA operands are read transposed from them; the next tile's rows are in flight (registers) meanwhile.
template <int H>
struct WgradGeom {
    static constexpr int NW = H >= 64 ? 16 : 8;               // waves
    static constexpr int NT = 64 * NW;
    static constexpr int TI = 3 * H / 16, TJ = 2 * H / 16;    // 16x16 output tiles
    static constexpr int ITW = 3, JTW = TJ / 4;               // per wave: 3 x JTW tiles; 4 wave groups across TJ
    static_assert(TI / ITW * 4 == NW && JTW >= 1, "wave grid must cover the output");
    static constexpr int LDG = 3 * H + 8, LDZ = 2 * H + 8;    // bf16 plane rows
    static constexpr int GPB = kTileRows * LDG * 2, ZPB = kTileRows * LDZ * 2;
    static constexpr int smem_bytes = 2 * GPB + 2 * ZPB;
    static constexpr int F4G = kTileRows * 3 * H / 4, F4Z = kTileRows * 2 * H / 4;
    static constexpr int PF = (F4G + F4Z) / NT;               // float4 loads per thread per tile
    static_assert(PF * NT == F4G + F4Z, "row loads must split evenly");
};

template <int H>
__global__ __launch_bounds__(WgradGeom<H>::NT) void k_sweep_wgrad_x3(const float* dgrows, const float* zrows, const int32_t* tile_list,
                                                                     int ntiles, const int32_t* tile_start, const int32_t* tile_count,
                                                                     float* slab) {      // [gridDim][3H * 2H] per-workgroup partials
    using G = WgradGeom<H>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* g_hi = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* g_lo = reinterpret_cast<__bf16*>(smem_raw + G::GPB);
    __bf16* z_hi = reinterpret_cast<__bf16*>(smem_raw + 2 * G::GPB);
    __bf16* z_lo = reinterpret_cast<__bf16*>(smem_raw + 2 * G::GPB + G::ZPB);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int it0 = (w / 4) * G::ITW, jt0 = (w % 4) * G::JTW;
    f32x4 acc[G::ITW][G::JTW];
#pragma unroll
    for (int i = 0; i < G::ITW; ++i)
#pragma unroll
        for (int j = 0; j < G::JTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 pf[G::PF];
    auto prefetch = [&](int k) {
        const int t = tile_list[k];
        const int64_t start = tile_start[t];
        const int cnt = tile_count[t];
#pragma unroll
        for (int u = 0; u < G::PF; ++u) {
            const int f = tid + u * G::NT;
            const bool isg = f < G::F4G;
            const int fz = f - G::F4G;
            const int row = isg ? f / (3 * H / 4) : fz / (2 * H / 4);
            const float* src = isg ? dgrows + start * 3 * H + (int64_t)f * 4 : zrows + start * 2 * H + (int64_t)fz * 4;
            if (row < cnt) pf[u] = *reinterpret_cast<const f32x4*>(src);
            else pf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    int k = blockIdx.x;
    if (k < ntiles) prefetch(k);
    for (; k < ntiles; k += gridDim.x) {
#pragma unroll
        for (int u = 0; u < G::PF; ++u) {
            const int f = tid + u * G::NT;
            const bool isg = f < G::F4G;
            const int fz = f - G::F4G;
            const int row = isg ? f / (3 * H / 4) : fz / (2 * H / 4);
            const int c4 = isg ? f % (3 * H / 4) : fz % (2 * H / 4);
            bf16x4 hi, lo;
            split4(f4(pf[u]), hi, lo);
            if (isg) { st_bf4(g_hi + row * G::LDG + 4 * c4, hi); st_bf4(g_lo + row * G::LDG + 4 * c4, lo); }
            else { st_bf4(z_hi + row * G::LDZ + 4 * c4, hi); st_bf4(z_lo + row * G::LDZ + 4 * c4, lo); }
        }
        lds_barrier();
        if (k + (int)gridDim.x < ntiles) prefetch(k + gridDim.x);
#pragma unroll
        for (int ks = 0; ks < kTileRows / 32; ++ks) {
            bf16x8 bh[G::JTW], bl[G::JTW];
#pragma unroll
            for (int j = 0; j < G::JTW; ++j) {
                bh[j] = ldfrag_tr2(z_hi, G::LDZ, 32 * ks, (jt0 + j) * 16);
                bl[j] = ldfrag_tr2(z_lo, G::LDZ, 32 * ks, (jt0 + j) * 16);
            }
#pragma unroll
            for (int i = 0; i < G::ITW; ++i) {
                const bf16x8 ah = ldfrag_tr2(g_hi, G::LDG, 32 * ks, (it0 + i) * 16), al = ldfrag_tr2(g_lo, G::LDG, 32 * ks, (it0 + i) * 16);
#pragma unroll
                for (int j = 0; j < G::JTW; ++j) mma_x3(acc[i][j], ah, al, bh[j], bl[j]);
            }
        }
        lds_barrier();
    }
#pragma unroll
    for (int i = 0; i < G::ITW; ++i)
#pragma unroll
        for (int j = 0; j < G::JTW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                slab[(int64_t)blockIdx.x * (6 * H * H) + ((it0 + i) * 16 + q * 4 + e) * 2 * H + (jt0 + j) * 16 + r] = acc[i][j][e];
}

// d_attn_u, dbvc, dbih, dbhh += sums of the per-workgroup slabs, one thread per address, slabs in index order (four
// interleaved partial sums combined in a fixed order): deterministic
template <int H>
__global__ __launch_bounds__(256) void k_level_small_reduce(const float* wslab, int nslab, int T, float* d_attn_u, float* dbvc,
                                                           float* dbih, float* dbhh) {
    const int g = blockIdx.x;
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= 11 * H) return;
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    const float* src = wslab + (int64_t)g * (11 * H) + i;
    int b = 0;
    for (; b + 4 <= nslab; b += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] += src[(int64_t)(b + k) * T * (11 * H)];
    }
    for (int k = 0; b < nslab; ++b, ++k) p[k] += src[(int64_t)b * T * (11 * H)];
    float* dst = i < 2 * H ? d_attn_u + (int64_t)g * 2 * H + i
               : i < 5 * H ? dbvc + (int64_t)g * 3 * H + (i - 2 * H)
               : i < 8 * H ? dbih + (int64_t)g * 3 * H + (i - 5 * H) : dbhh + (int64_t)g * 3 * H + (i - 8 * H);
    *dst += (p[0] + p[1]) + (p[2] + p[3]);
}

template <int H, bool HID>
int launch_level_x3_v(bool bwd, const LevelX3Args& a, int ntiles, hipStream_t st) {
    using M = LvlSmem<H>;
    if (bwd) {
        static bool set_b = false;
        if (!set_b) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_level_bwd_x3<H, HID>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set_b = true; }
        hipLaunchKernelGGL((k_level_bwd_x3<H, HID>), dim3(ntiles), dim3(kLT), M::bwd_bytes, st, a);
    } else {
        static bool set_f = false;
        if (!set_f) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_level_fwd_x3<H, HID>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set_f = true; }
        hipLaunchKernelGGL((k_level_fwd_x3<H, HID>), dim3(ntiles), dim3(kLT), M::fwd_bytes, st, a);
    }
    MGV_LAUNCH_RET();
}

template <int H>
int launch_level_x3(bool bwd, const LevelX3Args& a, int ntiles, hipStream_t st) {
    return a.gh != nullptr ? launch_level_x3_v<H, true>(bwd, a, ntiles, st) : launch_level_x3_v<H, false>(bwd, a, ntiles, st);
}

template <int H>
int launch_sweep_wgrad(const LevelX3Args& a, const int32_t* tile_list, int ntiles, float* dW, float* wgrad_slab, hipStream_t st) {
    using G = WgradGeom<H>;
    static bool set = false;
    if (!set) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_wgrad_x3<H>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    const int grid = std::min(ntiles, kWgradGrid);
    hipLaunchKernelGGL(k_sweep_wgrad_x3<H>, dim3(grid), dim3(G::NT), G::smem_bytes, st, a.dgrows, a.zrows, tile_list, ntiles,
                       a.tile_start, a.tile_count, wgrad_slab);
    launch_slab_sum<float, float>(wgrad_slab, grid, 6 * H * H, 6 * H * H, dW, st);      // fixed order: deterministic
    MGV_LAUNCH_RET();
}

}  // namespace mgv

#ifdef MGV_STAMPS
static unsigned long long* g_lvl_stamps = nullptr;
extern "C" int mgv_diag_set_level_stamps(void* p) { g_lvl_stamps = static_cast<unsigned long long*>(p); return 0; }
#define MGV_SET_LVL_STAMPS(a) (a).stamps = g_lvl_stamps
#else
#define MGV_SET_LVL_STAMPS(a)
#endif

extern "C" int mgv_sweep_zero_inactive(int H, int64_t N, const uint8_t* gslot, float* hf, void* stream) {
    MGV_CHECK_ARG(N >= 0 && (H == 16 || H == 32 || H == 64) && (N == 0 || (gslot && hf)));
    if (N == 0) return MGV_OK;
    const int rows_per_block = mgv::kThreads / (H / 4);
    hipLaunchKernelGGL(mgv::k_zero_inactive_rows, dim3(mgv::grid_for((N + rows_per_block - 1) / rows_per_block, 8)), dim3(mgv::kThreads), 0,
                       static_cast<hipStream_t>(stream), N, H, gslot, hf);
    MGV_LAUNCH_RET();
}

static int sweep_fwd_x3_impl(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                             const int32_t* order, const int32_t* order_span, int order_span_ints, const int32_t* tile_start,
                             const int32_t* tile_count, const int32_t* tile_slot, const int32_t* in_ptr,
                             const int32_t* in_src, const float* hs, float* hf, const float* attn_u,
                             const void* wpack_bf16, const float* bvc, const float* bih, const float* bhh,
                             const float* gh, const float* h_prev, void* stream) {
    MGV_CHECK_ARG(N >= 0 && T >= 1 && T <= mgv::kMaxSlots && num_levels >= 0 && level_tile_ptr_host && hs && hf && attn_u && wpack_bf16 && bvc && bih && bhh && in_ptr);
    MGV_CHECK_ARG(order_span_ints == 4 || order_span_ints == mgv::kRowInts);
    mgv::LevelX3Args a{};
    MGV_SET_LVL_STAMPS(a);
    a.N = N; a.T = T; a.order = order; a.order_span = order_span; a.span_ints = order_span_ints; a.tile_start = tile_start; a.tile_count = tile_count; a.tile_slot = tile_slot;
    a.in_ptr = in_ptr; a.in_src = in_src; a.hs = hs; a.hf = hf; a.attn_u = attn_u; a.wpack = static_cast<const __bf16*>(wpack_bf16);
    a.bvc = bvc; a.bih = bih; a.bhh = bhh;
    MGV_CHECK_ARG((gh == nullptr) == (h_prev == nullptr));
    a.gh = gh; a.hprev = h_prev;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int lv = 1; lv < num_levels; ++lv) {
        const int t0 = level_tile_ptr_host[lv], t1 = level_tile_ptr_host[lv + 1];
        if (t1 <= t0) continue;
        MGV_CHECK_ARG(order && order_span && tile_start && tile_count && tile_slot && in_src);
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

extern "C" int mgv_func_sweep_fwd_x3(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                     const int32_t* order, const int32_t* order_span, int order_span_ints, const int32_t* tile_start,
                                     const int32_t* tile_count, const int32_t* tile_slot, const int32_t* in_ptr,
                                     const int32_t* in_src, const float* hs, float* hf, const float* attn_u,
                                     const void* wpack_bf16, const float* bvc, const float* bih, const float* bhh, void* stream) {
    return sweep_fwd_x3_impl(H, N, T, num_levels, level_tile_ptr_host, order, order_span, order_span_ints, tile_start, tile_count, tile_slot, in_ptr, in_src, hs, hf,
                             attn_u, wpack_bf16, bvc, bih, bhh, nullptr, nullptr, stream);
}

extern "C" int mgv_func_sweep_round_fwd_x3(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                           const int32_t* order, const int32_t* order_span, int order_span_ints, const int32_t* tile_start,
                                           const int32_t* tile_count, const int32_t* tile_slot, const int32_t* in_ptr,
                                           const int32_t* in_src, const float* hs, float* hf, const float* attn_u,
                                           const void* wpack_bf16, const float* bvc, const float* bih, const float* zero_bhh,
                                           const float* gh, const float* h_prev, void* stream) {
    MGV_CHECK_ARG(gh && h_prev);
    return sweep_fwd_x3_impl(H, N, T, num_levels, level_tile_ptr_host, order, order_span, order_span_ints, tile_start, tile_count, tile_slot, in_ptr, in_src, hs, hf,
                             attn_u, wpack_bf16, bvc, bih, zero_bhh, gh, h_prev, stream);
}

static int sweep_bwd_x3_impl(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                     const int32_t* order, const int32_t* order_span, int order_span_ints, int64_t n_active,
                                     const int32_t* tile_start, const int32_t* tile_count, const int32_t* tile_slot,
                                     const int32_t* slot_tiles, const int32_t* slot_tile_ptr_host, const int32_t* in_ptr,
                                     const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                     const int32_t* out_slot, const uint8_t* gslot, const float* hs, const float* hf,
                                     const float* attn_u, const void* wpack_bf16, const float* bvc, const float* bih,
                                     const float* bhh, const float* ghf, float* ghs, float* dzb, float* alpha, float* dsc,
                                     float* d_attn_u, float* dWvc, float* dbvc, float* dbih, float* dbhh, float* scratch,
                                     int64_t scratch_elems, int skip_inactive_longer_than, int heavy_active_n,
                                     const int32_t* heavy_nodes, const int32_t* heavy_node_seg_ptr, const int32_t* heavy_seg_e0,
                                     const int32_t* heavy_seg_e1, const int32_t* heavy_lvl_k_ptr_host,
                                     const int32_t* heavy_lvl_seg_ptr_host, float* heavy_ws, int skip_active_longer_than,
                                     const float* gh, const float* h_prev, float* d_gh, float* g_hprev, void* stream) {
    MGV_CHECK_ARG(N >= 0 && T >= 1 && T <= mgv::kMaxSlots && num_levels >= 0 && n_active >= 0 && level_tile_ptr_host && hs && hf &&
                  attn_u && wpack_bf16 && bvc && bih && bhh);
    MGV_CHECK_ARG(in_ptr && out_ptr && gslot && ghf && ghs && dzb && d_attn_u && dWvc && dbvc && dbih && dbhh);
    MGV_CHECK_ARG(order_span_ints == 4 || order_span_ints == mgv::kRowInts);
    if (N == 0) return MGV_OK;
    mgv::LevelX3Args a{};
    MGV_SET_LVL_STAMPS(a);
    a.N = N; a.T = T; a.order = order; a.order_span = order_span; a.span_ints = order_span_ints; a.tile_start = tile_start; a.tile_count = tile_count; a.tile_slot = tile_slot;
    a.in_ptr = in_ptr; a.in_src = in_src; a.hs = hs; a.hf = const_cast<float*>(hf); a.attn_u = attn_u;
    a.wpack = static_cast<const __bf16*>(wpack_bf16); a.bvc = bvc; a.bih = bih; a.bhh = bhh;
    a.out_ptr = out_ptr; a.out_dst = out_dst; a.out_slot = out_slot; a.gslot = gslot;
    a.ghf = ghf; a.ghs = ghs; a.dzb = dzb; a.alpha = alpha; a.dsc = dsc; a.d_attn_u = d_attn_u; a.dWvc = dWvc; a.dbvc = dbvc;
    a.dbih = dbih; a.dbhh = dbhh; a.skip_inactive = skip_inactive_longer_than;
    MGV_CHECK_ARG(gh == nullptr ? (!h_prev && !d_gh && !g_hprev) : (h_prev && d_gh && g_hprev));
    a.gh = gh; a.hprev = h_prev; a.dgh = d_gh; a.ghprev = g_hprev;
    MGV_CHECK_ARG(heavy_active_n >= 0 && (heavy_active_n == 0 || (heavy_nodes && heavy_node_seg_ptr && heavy_seg_e0 && heavy_seg_e1 &&
                                                                heavy_lvl_k_ptr_host && heavy_lvl_seg_ptr_host && heavy_ws && skip_active_longer_than > 0)));
    a.skip_active = skip_active_longer_than; a.heavy_nodes = heavy_nodes; a.heavy_pull = heavy_ws;      // [K][2H], then the segment partials
    hipStream_t st = static_cast<hipStream_t>(stream);
    int nslab = 0;
    for (int lv = 1; lv < num_levels; ++lv) nslab = std::max(nslab, level_tile_ptr_host[lv + 1] - level_tile_ptr_host[lv]);
    const int64_t slab_elems = (int64_t)nslab * T * 11 * H;
    float* wgrad_slab = nullptr;
    if (nslab > 0) {
        MGV_CHECK_ARG(scratch && scratch_elems >= n_active * 5 * H + slab_elems + (int64_t)mgv::kWgradGrid * 6 * H * H && slot_tiles && slot_tile_ptr_host);
        a.dgrows = scratch; a.zrows = scratch + n_active * 3 * H; a.wslab = a.zrows + n_active * 2 * H;
        wgrad_slab = a.wslab + slab_elems;
        const hipError_t e = hipMemsetAsync(a.wslab, 0, slab_elems * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    for (int lv = num_levels - 1; lv >= 1; --lv) {
        const int t0 = level_tile_ptr_host[lv], t1 = level_tile_ptr_host[lv + 1];
        if (t1 <= t0) continue;
        MGV_CHECK_ARG(order && order_span && tile_start && tile_count && tile_slot && in_src && out_dst && out_slot && alpha && dsc);
        a.tile_begin = t0;
        a.heavy_k0 = a.heavy_k1 = 0;
        if (heavy_active_n > 0 && heavy_lvl_k_ptr_host[lv + 1] > heavy_lvl_k_ptr_host[lv]) {
            // this level's gates with very long consumer lists: their pulls by whole workgroups, one per list segment, before the level kernel
            const int k0 = heavy_lvl_k_ptr_host[lv], k1 = heavy_lvl_k_ptr_host[lv + 1];
            const int s0 = heavy_lvl_seg_ptr_host[lv], s1 = heavy_lvl_seg_ptr_host[lv + 1];
            float* partial = heavy_ws + (int64_t)heavy_active_n * 2 * H;
            const int grid = (s1 - s0) < 4096 ? (s1 - s0) : 4096;
            if (H == 32) hipLaunchKernelGGL((mgv::k_pull_heavy_seg<32, true>), dim3(grid), dim3(256), 0, st, s0, s1 - s0, heavy_seg_e0, heavy_seg_e1, out_dst, out_slot, gslot, alpha, dsc, dzb, attn_u, partial);
            else hipLaunchKernelGGL((mgv::k_pull_heavy_seg<64, true>), dim3(grid), dim3(256), 0, st, s0, s1 - s0, heavy_seg_e0, heavy_seg_e1, out_dst, out_slot, gslot, alpha, dsc, dzb, attn_u, partial);
            hipLaunchKernelGGL(mgv::k_heavy_add_range, dim3(((k1 - k0) * 2 + 15) / 16), dim3(256), 0, st, k0, k1, 2 * H, heavy_node_seg_ptr, partial, heavy_ws);
            a.heavy_k0 = k0; a.heavy_k1 = k1;
        }
        int rc;
        switch (H) {
            case 32: rc = mgv::launch_level_x3<32>(true, a, t1 - t0, st); break;
            case 64: rc = mgv::launch_level_x3<64>(true, a, t1 - t0, st); break;
            default: return MGV_EUNSUPPORTED;
        }
        if (rc != MGV_OK) return rc;
    }
    if (nslab > 0) {
        switch (H) {
            case 32: hipLaunchKernelGGL(mgv::k_level_small_reduce<32>, dim3(T, (11 * 32 + 255) / 256), dim3(256), 0, st, a.wslab, nslab, T, d_attn_u, dbvc, dbih, dbhh); break;
            case 64: hipLaunchKernelGGL(mgv::k_level_small_reduce<64>, dim3(T, (11 * 64 + 255) / 256), dim3(256), 0, st, a.wslab, nslab, T, d_attn_u, dbvc, dbih, dbhh); break;
            default: return MGV_EUNSUPPORTED;
        }
        for (int g = 0; g < T; ++g) {
            const int nt = slot_tile_ptr_host[g + 1] - slot_tile_ptr_host[g];
            if (nt <= 0) continue;
            const int rc = H == 32 ? mgv::launch_sweep_wgrad<32>(a, slot_tiles + slot_tile_ptr_host[g], nt, dWvc + (int64_t)g * 3 * H * 2 * H, wgrad_slab, st)
                                   : mgv::launch_sweep_wgrad<64>(a, slot_tiles + slot_tile_ptr_host[g], nt, dWvc + (int64_t)g * 3 * H * 2 * H, wgrad_slab, st);
            if (rc != MGV_OK) return rc;
        }
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

extern "C" int mgv_func_sweep_bwd_x3(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                     const int32_t* order, const int32_t* order_span, int order_span_ints, int64_t n_active,
                                     const int32_t* tile_start, const int32_t* tile_count, const int32_t* tile_slot,
                                     const int32_t* slot_tiles, const int32_t* slot_tile_ptr_host, const int32_t* in_ptr,
                                     const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                     const int32_t* out_slot, const uint8_t* gslot, const float* hs, const float* hf,
                                     const float* attn_u, const void* wpack_bf16, const float* bvc, const float* bih,
                                     const float* bhh, const float* ghf, float* ghs, float* dzb, float* alpha, float* dsc,
                                     float* d_attn_u, float* dWvc, float* dbvc, float* dbih, float* dbhh, float* scratch,
                                     int64_t scratch_elems, int skip_inactive_longer_than, int heavy_active_n,
                                     const int32_t* heavy_nodes, const int32_t* heavy_node_seg_ptr, const int32_t* heavy_seg_e0,
                                     const int32_t* heavy_seg_e1, const int32_t* heavy_lvl_k_ptr_host,
                                     const int32_t* heavy_lvl_seg_ptr_host, float* heavy_ws, int skip_active_longer_than, void* stream) {
    return sweep_bwd_x3_impl(H, N, T, num_levels, level_tile_ptr_host, order, order_span, order_span_ints, n_active, tile_start, tile_count, tile_slot, slot_tiles,
                             slot_tile_ptr_host, in_ptr, in_src, out_ptr, out_dst, out_slot, gslot, hs, hf, attn_u, wpack_bf16, bvc, bih, bhh, ghf, ghs, dzb,
                             alpha, dsc, d_attn_u, dWvc, dbvc, dbih, dbhh, scratch, scratch_elems, skip_inactive_longer_than, heavy_active_n, heavy_nodes,
                             heavy_node_seg_ptr, heavy_seg_e0, heavy_seg_e1, heavy_lvl_k_ptr_host, heavy_lvl_seg_ptr_host, heavy_ws, skip_active_longer_than, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int mgv_func_sweep_round_bwd_x3(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                                     const int32_t* order, const int32_t* order_span, int order_span_ints, int64_t n_active,
                                     const int32_t* tile_start, const int32_t* tile_count, const int32_t* tile_slot,
                                     const int32_t* slot_tiles, const int32_t* slot_tile_ptr_host, const int32_t* in_ptr,
                                     const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                     const int32_t* out_slot, const uint8_t* gslot, const float* hs, const float* hf,
                                     const float* attn_u, const void* wpack_bf16, const float* bvc, const float* bih,
                                     const float* bhh, const float* ghf, float* ghs, float* dzb, float* alpha, float* dsc,
                                     float* d_attn_u, float* dWvc, float* dbvc, float* dbih, float* dbhh, float* scratch,
                                     int64_t scratch_elems, int skip_inactive_longer_than, int heavy_active_n,
                                     const int32_t* heavy_nodes, const int32_t* heavy_node_seg_ptr, const int32_t* heavy_seg_e0,
                                     const int32_t* heavy_seg_e1, const int32_t* heavy_lvl_k_ptr_host,
                                     const int32_t* heavy_lvl_seg_ptr_host, float* heavy_ws, int skip_active_longer_than,
                                     const float* gh, const float* h_prev, float* d_gh, float* g_hprev, void* stream) {
    MGV_CHECK_ARG(gh && h_prev && d_gh && g_hprev);
    return sweep_bwd_x3_impl(H, N, T, num_levels, level_tile_ptr_host, order, order_span, order_span_ints, n_active, tile_start, tile_count, tile_slot, slot_tiles,
                             slot_tile_ptr_host, in_ptr, in_src, out_ptr, out_dst, out_slot, gslot, hs, hf, attn_u, wpack_bf16, bvc, bih, bhh, ghf, ghs, dzb,
                             alpha, dsc, d_attn_u, dWvc, dbvc, dbih, dbhh, scratch, scratch_elems, skip_inactive_longer_than, heavy_active_n, heavy_nodes,
                             heavy_node_seg_ptr, heavy_seg_e0, heavy_seg_e1, heavy_lvl_k_ptr_host, heavy_lvl_seg_ptr_host, heavy_ws, skip_active_longer_than, gh, h_prev, d_gh, g_hprev, stream);
}

// ghs[nodes[k]] = the pull of heavy never-updated nodes (skipped by mgv_func_sweep_bwd_x3 when skip_inactive_longer_than > 0), their
// consumer lists cut into segments (GraphPlan.heavy_segments), one workgroup per segment, partials added in segment order
extern "C" int mgv_sweep_pull_heavy(int H, int K, const int32_t* nodes, const int32_t* node_seg_ptr, int S, const int32_t* seg_e0,
                                    const int32_t* seg_e1, const int32_t* out_dst, const int32_t* out_slot, const uint8_t* gslot,
                                    const float* alpha, const float* dsc, const float* dzb, const float* attn_u, float* partial_ws,
                                    float* ghs, void* stream) {
    MGV_CHECK_ARG(K >= 0 && S >= 0 && ghs);
    if (K == 0 || S == 0) return MGV_OK;
    MGV_CHECK_ARG(nodes && node_seg_ptr && seg_e0 && seg_e1 && out_dst && out_slot && gslot && alpha && dsc && dzb && attn_u && partial_ws);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = S < 4096 ? S : 4096;
    switch (H) {
        case 32: hipLaunchKernelGGL((mgv::k_pull_heavy_seg<32, false>), dim3(grid), dim3(256), 0, st, 0, S, seg_e0, seg_e1, out_dst, out_slot, gslot, alpha, dsc, dzb, attn_u, partial_ws); break;
        case 64: hipLaunchKernelGGL((mgv::k_pull_heavy_seg<64, false>), dim3(grid), dim3(256), 0, st, 0, S, seg_e0, seg_e1, out_dst, out_slot, gslot, alpha, dsc, dzb, attn_u, partial_ws); break;
        default: return MGV_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(mgv::k_heavy_add, dim3((K + 15) / 16 < 1024 ? (K + 15) / 16 : 1024), dim3(256), 0, st, K, H, nodes, node_seg_ptr, partial_ws, ghs, H, 0);
    MGV_LAUNCH_RET();
}
