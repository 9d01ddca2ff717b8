// The levelised functional sweep (dg_ae_model_aig.py:70-97, arch/tfmlp.py:38-46) as ONE persistent kernel per direction
// instead of one launch per logic level (func_level_x3.hip).  bf16x3 arithmetic, H = 64, round 1 of the sweep (h0 = 0).
//
//   * one 8-wave workgroup per CU; every workgroup is DEDICATED to one aggregator slot (gate type) for the whole sweep: its slot's
//     Wvc hi / lo pack (96 KB in MFMA fragment order) is copied to LDS once and serves every tile of every level — no tile
//     re-fetches weights from L2;
//   * within a slot's workgroup set {j = 0 .. w-1}, workgroup j takes tiles j, j + w, ... of the (level, slot) tile range
//     (GraphPlan.key_tile_ptr), level after level; levels are separated by the XCD-hierarchical grid barrier of mgv_gridbar.h;
//   * what does not depend on the previous level is fetched BEFORE the barrier, one tile ahead: the next tile's descriptor, node ids
//     and CSR spans (order, order_span), its in-edge lists and — backward — its consumer lists with the consumers' slots; after
//     the barrier a tile is one round trip (the rows the previous level wrote) away from its arithmetic;
//   * rows another workgroup will read (hf forward; d(zbar), dL/dhs, alpha, d(score) backward) are stored write-through (sc1);
//   * backward: the weight gradient dWvc[slot] is accumulated in the workgroup's registers across ALL levels (48 VGPRs per wave) and
//     leaves once, through a per-workgroup slab row summed in workgroup order: the per-level kernels' 2.1 KB per node of gate-
//     gradient / zbar rows for a deferred weight-gradient kernel, and that kernel, do not exist here.  The small parameter gradients
//     (attention vector, biases) are register sums as well.  No float atomics: two identical calls give identical bits.
// The grid never exceeds the CU count (the launchers check it), every spin is bounded (mgv_gridbar.h) and a give-up is recorded in
// a sticky status word the host reads (mgv_sweep_persist_status).  Lists the per-level kernels treat by pre-passes (updated gates
// with more than 64 consumers), rounds >= 2 and other widths stay with the per-level kernels (ops.FuncSweepFn picks).
#include <cstdlib>
#include "func_level_x3_common.h"
#include "mgv_gridbar.h"

namespace mgv {

struct PersistArgs {
    LevelX3Args a;
    int num_levels;
    const int32_t* key_tile_ptr;      // [num_levels * T + 1]: tile range of every (level, slot) key
    int wg_begin[kMaxSlots + 1];      // workgroups [wg_begin[g], wg_begin[g + 1]) serve slot g; wg_begin[T] = grid
    GridBarState* bar;
    unsigned* sticky;                 // != 0 after a launch whose barrier gave up (never reset by the launchers)
    float* wg_slab;                   // backward: [grid][6 H^2 + 11 H] per-workgroup gradient partials
};

// this workgroup's walk over its tiles: (level, slot) ranges strided by the slot's workgroup count; wave-uniform
struct TileCursor {
    int lv, t, end;
    __device__ __forceinline__ void skip_empty(const PersistArgs& pa, int g, int j) {
        while (lv < pa.num_levels && t >= end) {
            ++lv;
            if (lv < pa.num_levels) { t = pa.key_tile_ptr[lv * pa.a.T + g] + j; end = pa.key_tile_ptr[lv * pa.a.T + g + 1]; }
        }
    }
    __device__ __forceinline__ void first(const PersistArgs& pa, int g, int j) {
        lv = 1; t = 0; end = 0;
        if (lv < pa.num_levels) { t = pa.key_tile_ptr[lv * pa.a.T + g] + j; end = pa.key_tile_ptr[lv * pa.a.T + g + 1]; }
        skip_empty(pa, g, j);
    }
    __device__ __forceinline__ void advance(const PersistArgs& pa, int g, int j, int w) { t += w; skip_empty(pa, g, j); }
    __device__ __forceinline__ bool valid(const PersistArgs& pa) const { return lv < pa.num_levels; }
};
// the same walk from the LAST level down (backward sweep)
struct TileCursorDown {
    int lv, t, end;
    __device__ __forceinline__ void skip_empty(const PersistArgs& pa, int g, int j) {
        while (lv >= 1 && t >= end) {
            --lv;
            if (lv >= 1) { t = pa.key_tile_ptr[lv * pa.a.T + g] + j; end = pa.key_tile_ptr[lv * pa.a.T + g + 1]; }
        }
    }
    __device__ __forceinline__ void first(const PersistArgs& pa, int g, int j) {
        lv = pa.num_levels - 1; t = 0; end = 0;
        if (lv >= 1) { t = pa.key_tile_ptr[lv * pa.a.T + g] + j; end = pa.key_tile_ptr[lv * pa.a.T + g + 1]; }
        skip_empty(pa, g, j);
    }
    __device__ __forceinline__ void advance(const PersistArgs& pa, int g, int j, int w) { t += w; skip_empty(pa, g, j); }
    __device__ __forceinline__ bool valid(const PersistArgs&) const { return lv >= 1; }
};

template <int H>
struct PFwdSmem {
    using M = LvlSmem<H>;
    static constexpr int WB = 2 * 6 * H * H * 2;             // Wvc hi | lo in fragment order
    static constexpr int o_w = 0;
    static constexpr int o_zhi = o_w + WB;
    static constexpr int o_zlo = o_zhi + M::ZPB;
    static_assert(M::DHB <= 2 * M::ZPB, "forward output tile reuses the zbar planes");
    static constexpr int o_sa = o_zlo + M::ZPB;              // sa[64], the slot's attention vector u[2H] and bias rows bvc | bih | bhh [3H each]
    static constexpr int o_idx = o_sa + (kTileRows + 2 * H + 9 * H) * 4;       // two index sets: this tile's and the next one's
    static constexpr int o_flag = o_idx + 2 * M::IDX_B;
    static constexpr int bytes = o_flag + 16;
    static_assert(bytes <= 160 * 1024, "LDS");
};

// workgroup's slot g, its index j among the slot's w workgroups
__device__ __forceinline__ void wg_role(const PersistArgs& pa, int& g, int& j, int& w) {
    g = 0;
    while (g + 1 < pa.a.T && (int)blockIdx.x >= pa.wg_begin[g + 1]) ++g;
    j = (int)blockIdx.x - pa.wg_begin[g];
    w = pa.wg_begin[g + 1] - pa.wg_begin[g];
}

__device__ __forceinline__ void copy_to_lds16(unsigned char* dst, const void* src, int bytes) {
    const float4* s = reinterpret_cast<const float4*>(src);
    float4* d = reinterpret_cast<float4*>(dst);
    for (int i = threadIdx.x; i < bytes / 16; i += kLT) d[i] = s[i];
}

template <int H>
__global__ __launch_bounds__(kLT, 4) void k_sweep_fwd_persist(PersistArgs pa) {
    using S = SplitL<H>;
    using M = LvlSmem<H>;
    using P = PFwdSmem<H>;
    constexpr int LPR = M::LPR, GROUPS = M::GROUPS, RPG = M::RPG, LDO = M::LDO, BLK = 6 * H * H;
    const LevelX3Args& a = pa.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const __bf16* w_lds = reinterpret_cast<const __bf16*>(smem_raw + P::o_w);
    __bf16* z_hi = reinterpret_cast<__bf16*>(smem_raw + P::o_zhi);
    __bf16* z_lo = reinterpret_cast<__bf16*>(smem_raw + P::o_zlo);
    float* s_o = reinterpret_cast<float*>(smem_raw + P::o_zhi);
    float* s_sa = reinterpret_cast<float*>(smem_raw + P::o_sa);
    int* s_flag = reinterpret_cast<int*>(smem_raw + P::o_flag);
    int g, j, w;
    wg_role(pa, g, j, w);
    const int tid = threadIdx.x;
    copy_to_lds16(smem_raw + P::o_w, a.wpack + (int64_t)g * 4 * BLK, P::WB);
    // slot constants in LDS: the attention vector, the three bias rows
    float* s_u = s_sa + kTileRows;
    float* s_b = s_u + 2 * H;
    for (int i = tid; i < 2 * H; i += kLT) s_u[i] = a.attn_u[(int64_t)g * 2 * H + i];
    for (int i = tid; i < 3 * H; i += kLT) {
        s_b[i] = a.bvc[(int64_t)g * 3 * H + i]; s_b[3 * H + i] = a.bih[(int64_t)g * 3 * H + i]; s_b[6 * H + i] = a.bhh[(int64_t)g * 3 * H + i];
    }
    const __amdgpu_buffer_rsrc_t rs_hf = gb_rsrc(a.hf, (uint64_t)a.N * H * 4);
    GridBarLocal gb;
    if (!grid_barrier_init(pa.bar, gridDim.x, gb, s_flag)) { if (tid == 0) gb_store(pa.sticky, 1u); return; }

    TileCursor cur, nxt;
    cur.first(pa, g, j);
    nxt = cur;
    if (cur.valid(pa)) nxt.advance(pa, g, j, w);
    int cur_start = 0, cur_count = 0, nxt_start = 0, nxt_count = 0;
    int p = 0;
    if (cur.valid(pa)) {
        // the first tile's index set, staged in place (every later one is staged one tile ahead, below)
        cur_start = a.tile_start[cur.t]; cur_count = a.tile_count[cur.t];
        stage_spans(a, cur_start, cur_count, lvl_idx(smem_raw + P::o_idx));
        lds_barrier();
        stage_in_edges(a, lvl_idx(smem_raw + P::o_idx));
    }
    if (nxt.valid(pa)) { nxt_start = a.tile_start[nxt.t]; nxt_count = a.tile_count[nxt.t]; }
    lds_barrier();

    for (int lvl = 1; lvl < pa.num_levels; ++lvl) {
        while (cur.valid(pa) && cur.lv == lvl) {
            const LvlIdx ix = lvl_idx(smem_raw + P::o_idx + p * M::IDX_B);
            const LvlIdx nx = lvl_idx(smem_raw + P::o_idx + (p ^ 1) * M::IDX_B);
            const bool have_next = nxt.valid(pa);
            // ---- F1: source rows (the only loads that wait for the previous level), attention, zbar planes
            // (every phase re-derives its lane constants behind an opaque copy of the thread id: hipcc otherwise hoists the LDS
            // addresses of all phases out of the persistent loop and spills them)
            {
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                const int grp = tid / LPR, lr = tid % LPR;
                const float4 us = ld4(s_u + 4 * lr), uf = ld4(s_u + H + 4 * lr);
                InRows<H> Lr[RPG];
                int4 sp[RPG];
#pragma unroll
                for (int i = 0; i < RPG; ++i) {
                    const int row = grp + i * GROUPS;
                    sp[i] = ix.span[row];
                    Lr[i].issue(a, ix.insrc + row * kInCap, sp[i].y - sp[i].x, lr);
                }
                // the NEXT tile's node ids and CSR spans ride behind the rows
                int n_node = -1;
                int4 n_sp = make_int4(0, 0, 0, 0);
                if (have_next && tid < kTileRows && tid < nxt_count) {
                    n_node = a.order[nxt_start + tid];
                    n_sp = *reinterpret_cast<const int4*>(a.order_span + 4 * (int64_t)(nxt_start + tid));
                }
#pragma unroll
                for (int i = 0; i < RPG; ++i) {
                    const int row = grp + i * GROUPS;
                    float m, inv;
                    float4 zs, zf;
                    attn_reduce<H>(a, Lr[i], sp[i], us, uf, lr, m, inv, zs, zf);
                    store_zbar<H>(z_hi, z_lo, row, lr, zs, zf);
                    if (lr == 0) s_sa[row] = sp[i].y > sp[i].x ? 1.0f : 0.0f;
                }
                if (tid < kTileRows) { nx.node[tid] = n_node; nx.span[tid] = n_sp; }
            }
            lds_barrier();
            // ---- F2: the next tile's in-edge lists (in flight across the MFMAs), gate pre-activations from the LDS-resident weights
            int n_in = 0;
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            if (have_next && tid < kTileRows * kInCap) {
                const int4 spn = nx.span[tid / kInCap];
                if (spn.x + tid % kInCap < spn.y) n_in = a.in_src[spn.x + tid % kInCap];
            }
            f32x4 ar[S::RTW], az[S::RTW], an[S::RTW];
            lvl_gemm_x3<H>(w_lds, z_hi, z_lo, ar, az, an);
            lds_barrier();                   // s_o overlays the planes
            // ---- F3: GRU from a zero state
            tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
            const int wc = wv % S::WPC, wr = wv / S::WPC, col = wc * 16 + r;
            const float bvr = s_b[col], bvz = s_b[H + col], bvn = s_b[2 * H + col];
            const float cr = s_b[3 * H + col] + s_b[6 * H + col], cz = s_b[4 * H + col] + s_b[7 * H + col], cn = s_b[5 * H + col];
            const float bhn = s_b[8 * H + col];
#pragma unroll
            for (int i = 0; i < S::RTW; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                    const float sa = s_sa[row];
                    const float rr = sigmoidf_(ar[i][e] + sa * bvr + cr);
                    const float zz = sigmoidf_(az[i][e] + sa * bvz + cz);
                    const float nn = tanhf_(an[i][e] + sa * bvn + cn + rr * bhn);
                    s_o[row * LDO + col] = (1.0f - zz) * nn;        // h0 = 0
                }
            if (tid < kTileRows * kInCap) nx.insrc[tid] = n_in;
            lds_barrier();
            // ---- F4: rows out, write-through: other workgroups read them behind the next grid barrier
            tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int grp = tid / LPR, lr = tid % LPR;
#pragma unroll
            for (int i = 0; i < RPG; ++i) {
                const int row = grp + i * GROUPS;
                if (row < cur_count) st4_wt(rs_hf, ((uint64_t)ix.node[row] * H + 4 * lr) * 4, ld4(s_o + row * LDO + 4 * lr));
            }
            lds_barrier();                   // the next tile's planes overlay s_o
            cur = nxt; cur_start = nxt_start; cur_count = nxt_count; p ^= 1;
            if (nxt.valid(pa)) {
                nxt.advance(pa, g, j, w);
                if (nxt.valid(pa)) { nxt_start = a.tile_start[nxt.t]; nxt_count = a.tile_count[nxt.t]; }
            }
        }
        if (lvl + 1 < pa.num_levels) {
            if (!grid_barrier<false>(pa.bar, gb, s_flag)) { if (tid == 0) gb_store(pa.sticky, 2u); return; }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int kOutCapP = 8;            // consumers per row staged in LDS (longer lists continue from global memory)

template <int H>
struct PBwdSmem {
    using M = LvlSmem<H>;
    static constexpr int WB = 2 * 6 * H * H * 2;             // Wvc hi | lo in fragment order (recompute)
    static constexpr int o_w = 0;
    static constexpr int o_zhi = o_w + WB;                    // zbar planes; after the three passes: d(zbar) fp32 [64][2H + 4]
    static constexpr int o_zlo = o_zhi + M::ZPB;
    static_assert(kTileRows * M::LDZF * 4 <= 2 * M::ZPB, "d(zbar) tile overlays the zbar planes");
    static constexpr int OUT_B = kTileRows * kOutCapP * (4 + 4 + 1);          // consumer, its in-CSR slot, its gate slot
    static constexpr int o_r = o_zlo + M::ZPB;                // region R: dh fp32 | staged consumers; in the passes: one gate's gradient planes
    static constexpr int o_out = o_r + M::DHB;
    static constexpr int R_BYTES = (M::DHB + OUT_B > 2 * M::GPB) ? M::DHB + OUT_B : 2 * M::GPB;
    static constexpr int o_small = o_r + R_BYTES;             // u of every slot [kMaxSlots][2H], sa / m / inv [64 each]
    static constexpr int SMALL_B = (kMaxSlots * 2 * H + 3 * kTileRows) * 4;
    static constexpr int o_idx = o_small + SMALL_B;           // two index sets: this tile's and the next one's
    static constexpr int o_flag = o_idx + 2 * M::IDX_B;
    static constexpr int bytes = o_flag + 16;
    static_assert(bytes <= 160 * 1024, "LDS");
    // end-of-kernel staging of the small sums overlays the planes: [2 row-tile waves][9H] bias sums, [8 waves][2H] attention vector
    static_assert((2 * 9 * H + kLW * 2 * H) * 4 <= 2 * M::ZPB, "final staging");
};

struct OutStageP { int* c; int* sl; uint8_t* gc; };
__device__ __forceinline__ OutStageP out_stage_p(unsigned char* base) {
    OutStageP o;
    o.c = reinterpret_cast<int*>(base);
    o.sl = reinterpret_cast<int*>(base + kTileRows * kOutCapP * 4);
    o.gc = reinterpret_cast<uint8_t*>(base + kTileRows * kOutCapP * 8);
    return o;
}

// consumers [k0, k0 + kOutChunk) of a row's staged list: their d(zbar) rows and the two per-edge scalars, all in flight together
template <int H>
struct OutRowsP {
    f32x4 ds_[kOutChunk], df_[kOutChunk];
    float al[kOutChunk], sc[kOutChunk];
    __device__ __forceinline__ void issue(const LevelX3Args& a, const OutStageP& o, int base, int k0, int n, int lr) {
#pragma unroll
        for (int k = 0; k < kOutChunk; ++k)
            if (k0 + k < n && o.gc[base + k0 + k] != kNoGateX) {
                const float* dz = a.dzb + (int64_t)o.c[base + k0 + k] * 2 * H;
                const int sl = o.sl[base + k0 + k];
                ds_[k] = *reinterpret_cast<const f32x4*>(dz + 4 * lr);
                df_[k] = *reinterpret_cast<const f32x4*>(dz + H + 4 * lr);
                al[k] = a.alpha[sl]; sc[k] = a.dsc[sl];
            }
    }
    __device__ __forceinline__ void reduce(const OutStageP& o, const float* uall, int base, int k0, int n, int lr, float4& gs, float4& gf) const {
#pragma unroll
        for (int k = 0; k < kOutChunk; ++k)
            if (k0 + k < n && o.gc[base + k0 + k] != kNoGateX) {
                const float* u = uall + (int)o.gc[base + k0 + k] * 2 * H;
                gs = fma4(al[k], f4(ds_[k]), fma4(sc[k], ld4(u + 4 * lr), gs));
                gf = fma4(al[k], f4(df_[k]), fma4(sc[k], ld4(u + H + 4 * lr), gf));
            }
    }
};

constexpr int kPBW = 12;               // waves of a backward workgroup: 8 row waves (0-7) + 4 weight-gradient waves (8-11), three per SIMD
constexpr int kPBT = 64 * kPBW;

// The backward workgroup.  Waves 0-7 ("row waves") do what a per-level backward workgroup does; waves 8-11 hold the slot's weight
// gradient dWvc[3H][2H] in registers for the whole sweep (wave 8 + i: rows 16 i .. 16 i + 15 of every gate block, 96 accumulator
// VGPRs) and add a tile's contribution in the three gate passes, beside the row waves' dgrad MFMAs: both read the same gate-
// gradient planes, the weight-gradient waves transposed, and the zbar planes transposed.  All twelve waves run the same barrier
// sequence.  (With the accumulators on the row waves, 48 VGPRs each, hipcc spilled them around every other phase.)
template <int H, bool WGW>      // WGW: four extra weight-gradient waves (else the row waves hold the accumulators, 48 VGPRs each)
__global__ __launch_bounds__(WGW ? kPBT : kLT, WGW ? 3 : 2) void k_sweep_bwd_persist(PersistArgs pa) {
    using S = SplitL<H>;
    using S2 = SplitL<2 * H>;
    using M = LvlSmem<H>;
    using P = PBwdSmem<H>;
    static_assert(H == 64, "wave roles below are laid out for H = 64");
    constexpr int LPR = M::LPR, GROUPS = M::GROUPS, RPG = M::RPG, LDO = M::LDO;
    constexpr int LDZP = M::LDZP, LDGP = M::LDGP, LDZF = M::LDZF, BLK = 6 * H * H;
    static_assert(RPG == 2, "two rows per lane group");
    const LevelX3Args& a = pa.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const __bf16* w_lds = reinterpret_cast<const __bf16*>(smem_raw + P::o_w);
    __bf16* z_hi = reinterpret_cast<__bf16*>(smem_raw + P::o_zhi);
    __bf16* z_lo = reinterpret_cast<__bf16*>(smem_raw + P::o_zlo);
    float* s_dz = reinterpret_cast<float*>(smem_raw + P::o_zhi);
    float* s_dh = reinterpret_cast<float*>(smem_raw + P::o_r);
    __bf16* d_hi = reinterpret_cast<__bf16*>(smem_raw + P::o_r);
    __bf16* d_lo = d_hi + kTileRows * LDGP;
    const OutStageP os = out_stage_p(smem_raw + P::o_out);
    float* s_uall = reinterpret_cast<float*>(smem_raw + P::o_small);
    float* s_sa = s_uall + kMaxSlots * 2 * H;
    float* s_m = s_sa + kTileRows;
    float* s_inv = s_m + kTileRows;
    int* s_flag = reinterpret_cast<int*>(smem_raw + P::o_flag);
    int g, j, w;
    wg_role(pa, g, j, w);
    const bool row_wave = !WGW || __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) < kLW;       // wave-uniform
    {
        const float4* src = reinterpret_cast<const float4*>(a.wpack + (int64_t)g * 4 * BLK);
        float4* dst = reinterpret_cast<float4*>(smem_raw + P::o_w);
        for (int i = threadIdx.x; i < P::WB / 16; i += (WGW ? kPBT : kLT)) dst[i] = src[i];
        for (int i = threadIdx.x; i < a.T * 2 * H; i += (WGW ? kPBT : kLT)) s_uall[i] = a.attn_u[i];
    }
    if (WGW && !row_wave) {
    // ================= weight-gradient waves: the same cursor walk and the same barrier sequence as the row waves =================
    f32x4 wacc[3][2 * H / 16];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int jj = 0; jj < 2 * H / 16; ++jj) wacc[p][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    GridBarLocal gb;
    if (!grid_barrier_init(pa.bar, gridDim.x, gb, s_flag)) return;
    STAMP_DECL
    STAMP_BEGIN;
    TileCursorDown cur;
    cur.first(pa, g, j);
    if (cur.valid(pa)) lds_barrier();          // the first tile's staging
    lds_barrier();
    const int wi = (int)(threadIdx.x >> 6) - kLW;
    for (int lvl = pa.num_levels - 1; lvl >= 1; --lvl) {
        while (cur.valid(pa) && cur.lv == lvl) {
            lds_barrier();                     // A
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                lds_barrier();
                lds_barrier();                 // the gate's gradient planes are in LDS
                STAMP(0);
                // weight gradient of gate p: both operands read transposed from the row-major planes (k = the tile's 64 rows)
#pragma unroll
                for (int ks = 0; ks < kTileRows / 32; ++ks) {
                    const bf16x8 ah = ldfrag_tr2(d_hi, LDGP, 32 * ks, wi * 16), al = ldfrag_tr2(d_lo, LDGP, 32 * ks, wi * 16);
#pragma unroll
                    for (int jj = 0; jj < 2 * H / 16; ++jj) {
                        const bf16x8 bh = ldfrag_tr2(z_hi, LDZP, 32 * ks, jj * 16), bl = ldfrag_tr2(z_lo, LDZP, 32 * ks, jj * 16);
                        mma_x3(wacc[p][jj], ah, al, bh, bl);
                    }
                }
                STAMP(1);
            }
            lds_barrier();                     // E
            lds_barrier();
            lds_barrier();                     // F
            cur.advance(pa, g, j, w);
        }
        if (lvl > 1) {
            if (!grid_barrier<false>(pa.bar, gb, s_flag)) return;
        }
        STAMP(8);
    }
    STAMP_FLUSH(a);
    float* slab = pa.wg_slab + (int64_t)blockIdx.x * (6 * H * H + 11 * H);
    lds_barrier();
    {
        const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int jj = 0; jj < 2 * H / 16; ++jj)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    slab[(p * H + wi * 16 + q * 4 + e) * 2 * H + jj * 16 + r] = wacc[p][jj][e];
    }
    lds_barrier();
        return;
    }
    // ================= row waves =================
    // row waves: this lane's gate column's bias combinations, kept for the whole sweep
    float bvr = 0.f, bvz = 0.f, bvn = 0.f, cr = 0.f, cz = 0.f, cn = 0.f, bhn = 0.f;
    {
        const int col = ((threadIdx.x >> 6) % S::WPC) * 16 + (threadIdx.x & 15);
        const float* bvc = a.bvc + (int64_t)g * 3 * H; const float* bih = a.bih + (int64_t)g * 3 * H; const float* bhh = a.bhh + (int64_t)g * 3 * H;
        bvr = bvc[col]; bvz = bvc[H + col]; bvn = bvc[2 * H + col];
        cr = bih[col] + bhh[col]; cz = bih[H + col] + bhh[H + col]; cn = bih[2 * H + col]; bhn = bhh[2 * H + col];
    }
    // without weight-gradient waves: wave (i = wv & 3, jh = wv >> 2) owns rows 16 i .. 16 i + 15 of every gate block and columns 64 jh .. + 63 of dWvc
    f32x4 wacc[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) wacc[p][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    // row waves: small parameter gradients, per-lane partial sums over all this lane's rows of all tiles
    float b_r = 0.f, b_z = 0.f, b_n = 0.f, v_r = 0.f, v_z = 0.f, v_n = 0.f, h_n = 0.f;
    float4 gus = zero4(), guf = zero4();
    const __amdgpu_buffer_rsrc_t rs_dzb = gb_rsrc(a.dzb, (uint64_t)a.N * 2 * H * 4);
    GridBarLocal gb;
    if (!grid_barrier_init(pa.bar, gridDim.x, gb, s_flag)) { if (threadIdx.x == 0) gb_store(pa.sticky, 3u); return; }

    STAMP_DECL
    STAMP_BEGIN;
    TileCursorDown cur, nxt;
    cur.first(pa, g, j);
    nxt = cur;
    if (cur.valid(pa)) nxt.advance(pa, g, j, w);
    int cur_start = 0, cur_count = 0, nxt_start = 0, nxt_count = 0;
    int pp = 0;
    if (cur.valid(pa)) {
        // the first tile's index set and consumer lists, staged in place (every later one is staged one tile ahead)
        cur_start = a.tile_start[cur.t]; cur_count = a.tile_count[cur.t];
        const LvlIdx ix0 = lvl_idx(smem_raw + P::o_idx);
        stage_spans(a, cur_start, cur_count, ix0);
        lds_barrier();
        {
            stage_in_edges(a, ix0);
            const int row = threadIdx.x / kOutCapP, k = threadIdx.x % kOutCapP;
            const int4 sp = ix0.span[row];
            if (sp.z + k < sp.w) {
                const int c = a.out_dst[sp.z + k];
                os.c[threadIdx.x] = c; os.sl[threadIdx.x] = a.out_slot[sp.z + k]; os.gc[threadIdx.x] = a.gslot[c];
            }
        }
    }
    if (nxt.valid(pa)) { nxt_start = a.tile_start[nxt.t]; nxt_count = a.tile_count[nxt.t]; }
    lds_barrier();

    for (int lvl = pa.num_levels - 1; lvl >= 1; --lvl) {
        while (cur.valid(pa) && cur.lv == lvl) {
            const LvlIdx ix = lvl_idx(smem_raw + P::o_idx + pp * M::IDX_B);
            const LvlIdx nx = lvl_idx(smem_raw + P::o_idx + (pp ^ 1) * M::IDX_B);
            const bool have_next = nxt.valid(pa);
            // ---- A: pull dL/dhf, dL/dhs of the tile's nodes from their consumers (the loads that wait for the level above), recompute
            //         the attention; both rows of a lane group have all their loads in flight together
            {
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                const int grp = tid / LPR, lr = tid % LPR;
                const float4 us = ld4(s_uall + g * 2 * H + 4 * lr), uf = ld4(s_uall + g * 2 * H + H + 4 * lr);
                // the NEXT tile's node ids and CSR spans ride behind the rows
                int n_node = -1;
                int4 n_sp = make_int4(0, 0, 0, 0);
                if (have_next && tid < kTileRows && tid < nxt_count) {
                    n_node = a.order[nxt_start + tid];
                    n_sp = *reinterpret_cast<const int4*>(a.order_span + 4 * (int64_t)(nxt_start + tid));
                }
#pragma unroll 1
                for (int i = 0; i < RPG; ++i) {
                    const int row = grp + i * GROUPS;
                    const int4 sp = ix.span[row];
                    const int nout = min(sp.w - sp.z, kOutCapP);
                    InRows<H> Lr;
                    OutRowsP<H> Pr;
                    f32x4 own = f32x4{0.f, 0.f, 0.f, 0.f};
                    Pr.issue(a, os, row * kOutCapP, 0, nout, lr);
                    Lr.issue(a, ix.insrc + row * kInCap, sp.y - sp.x, lr);
                    if (ix.node[row] >= 0) own = *reinterpret_cast<const f32x4*>(a.ghf + (int64_t)ix.node[row] * H + 4 * lr);
                    float4 gs = zero4(), gf = zero4();
                    Pr.reduce(os, s_uall, row * kOutCapP, 0, nout, lr, gs, gf);
                    for (int k0 = kOutChunk; k0 < nout; k0 += kOutChunk) {
                        OutRowsP<H> Q;
                        Q.issue(a, os, row * kOutCapP, k0, nout, lr);
                        Q.reduce(os, s_uall, row * kOutCapP, k0, nout, lr, gs, gf);
                    }
                    pull_tail<H>(a, sp.z + kOutCapP, sp.w, lr, gs, gf);
                    const float4 dh = add4(gf, f4(own));
                    if (row < cur_count) st4(a.ghs + (int64_t)ix.node[row] * H + 4 * lr, gs);      // dL/dhs of the node: final, no other workgroup reads it
                    float m, inv;
                    float4 zs, zf;
                    attn_reduce<H>(a, Lr, sp, us, uf, lr, m, inv, zs, zf);
                    store_zbar<H>(z_hi, z_lo, row, lr, zs, zf);
                    st4(s_dh + row * LDO + 4 * lr, dh);
                    if (lr == 0) { s_sa[row] = sp.y > sp.x ? 1.0f : 0.0f; s_m[row] = m; s_inv[row] = inv; }
                }
                if (tid < kTileRows) { nx.node[tid] = n_node; nx.span[tid] = n_sp; }
            }
            STAMP(0);
            lds_barrier();
            STAMP(1);
            // ---- B: the next tile's edge lists (in flight across the MFMAs); recompute the gates from the LDS-resident weights
            // ---- C: GRU backward (h0 = 0: hf = (1 - z) n, gh = b_hh); ar / az / an become da_r / da_z / da_n
            int n_in = 0, n_c = 0, n_sl = 0, n_gc = kNoGateX;
            f32x4 ar[S::RTW], az[S::RTW], an[S::RTW];
            {
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                bool n_has = false;
                if (have_next) {
                    if (tid < kTileRows * kInCap) {
                        const int4 spn = nx.span[tid / kInCap];
                        if (spn.x + tid % kInCap < spn.y) n_in = a.in_src[spn.x + tid % kInCap];
                    }
                    const int4 spo = nx.span[tid / kOutCapP];
                    if (spo.z + tid % kOutCapP < spo.w) { n_has = true; n_c = a.out_dst[spo.z + tid % kOutCapP]; n_sl = a.out_slot[spo.z + tid % kOutCapP]; }
                }
                lvl_gemm_x3<H>(w_lds, z_hi, z_lo, ar, az, an);
                STAMP(2);
                if (n_has) n_gc = a.gslot[n_c];
                const int lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
                const int wc = wv % S::WPC, wr = wv / S::WPC, col = wc * 16 + r;
#pragma unroll
                for (int i = 0; i < S::RTW; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = (wr * S::RTW + i) * 16 + q * 4 + e;
                        const float sa = s_sa[row];
                        const float rr = sigmoidf_(ar[i][e] + sa * bvr + cr);
                        const float zz = sigmoidf_(az[i][e] + sa * bvz + cz);
                        const float nn = tanhf_(an[i][e] + sa * bvn + cn + rr * bhn);
                        const float dh = s_dh[row * LDO + col];
                        const float dan = dh * (1.0f - zz) * (1.0f - nn * nn);
                        const float daz = -dh * nn * zz * (1.0f - zz);
                        const float dar = dan * bhn * rr * (1.0f - rr);
                        ar[i][e] = dar; az[i][e] = daz; an[i][e] = dan;
                        b_r += dar; b_z += daz; b_n += dan; h_n += dan * rr;
                        v_r += sa * dar; v_z += sa * daz; v_n += sa * dan;
                    }
            }
            STAMP(3);
            // ---- D: three passes (r, z, n): the gate's gradient planes to LDS; row waves: d(zbar) += dG_p Wvc[p]; weight-gradient
            //         waves: dWvc[p] += dG_p^T zbar
            f32x4 dz[S2::RTW];
#pragma unroll
            for (int i = 0; i < S2::RTW; ++i) dz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                const int lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
                // dgrad B operands of this pass (WvcT hi / lo of this wave's column tile of d(zbar)): from L2, issued before the
                // barriers, their latency hides behind the plane writes
                bf16x8 tbh[H / 32], tbl[H / 32];
                {
                    const __bf16* wslot = a.wpack + (int64_t)g * 4 * BLK;
#pragma unroll
                    for (int ks = 0; ks < H / 32; ++ks) {
                        const int wo = (((wv % S2::WPC) * 3 + p) * (H / 32) + ks) * 512 + lane * 8;
                        tbh[ks] = ldfrag(wslot + 2 * BLK + wo); tbl[ks] = ldfrag(wslot + 3 * BLK + wo);
                    }
                }
                lds_barrier();                 // readers of region R: phase C (dh), or the previous pass
                {
                    const int wc = wv % S::WPC, wr = wv / S::WPC, col = wc * 16 + r;
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
                {
                    const int wr2 = wv / S2::WPC;
#pragma unroll
                    for (int ks = 0; ks < H / 32; ++ks)
#pragma unroll
                        for (int i = 0; i < S2::RTW; ++i) {
                            const int off = ((wr2 * S2::RTW + i) * 16 + r) * LDGP + 32 * ks + 8 * q;
                            mma_x3(dz[i], ldfrag(d_hi + off), ldfrag(d_lo + off), tbh[ks], tbl[ks]);
                        }
                }
                if constexpr (!WGW) {
                    // weight gradient of gate p: both operands read transposed from the row-major planes (k = the tile's 64 rows)
                    const int wi = wv & 3, jh = wv >> 2;
#pragma unroll
                    for (int ks = 0; ks < kTileRows / 32; ++ks) {
                        const bf16x8 ah = ldfrag_tr2(d_hi, LDGP, 32 * ks, wi * 16), al = ldfrag_tr2(d_lo, LDGP, 32 * ks, wi * 16);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const bf16x8 bh = ldfrag_tr2(z_hi, LDZP, 32 * ks, (jh * 4 + jj) * 16), bl = ldfrag_tr2(z_lo, LDZP, 32 * ks, (jh * 4 + jj) * 16);
                            mma_x3(wacc[p][jj], ah, al, bh, bl);
                        }
                    }
                }
            }
            STAMP(4);
            // ---- E: d(zbar) tile to LDS (fp32, row layout for the attention backward); it overlays the zbar planes
            lds_barrier();
            {
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                const int lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
                const int wc2 = wv % S2::WPC, wr2 = wv / S2::WPC, col = wc2 * 16 + r;
#pragma unroll
                for (int i = 0; i < S2::RTW; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s_dz[((wr2 * S2::RTW + i) * 16 + q * 4 + e) * LDZF + col] = dz[i][e];
                if (tid < kTileRows * kInCap) nx.insrc[tid] = n_in;
            }
            lds_barrier();
            STAMP(5);
            // ---- F: attention backward per in-edge; rows out (write-through: the levels below read them behind the next grid barrier)
            {
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                const int grp = tid / LPR, lr = tid % LPR;
                const float4 us = ld4(s_uall + g * 2 * H + 4 * lr), uf = ld4(s_uall + g * 2 * H + H + 4 * lr);
#pragma unroll 1
                for (int i = 0; i < RPG; ++i) {
                    const int row = grp + i * GROUPS;
                    const int4 sp = ix.span[row];
                    InRows<H> Lr;
                    Lr.issue(a, ix.insrc + row * kInCap, sp.y - sp.x, lr);
                    const float4 dzs = ld4(s_dz + row * LDZF + 4 * lr), dzf = ld4(s_dz + row * LDZF + H + 4 * lr);
                    float al[kInRegs], ds[kInRegs];
                    if (row < cur_count) {
                        attn_bwd_row<H, true>(a, Lr, sp, us, uf, dzs, dzf, s_m[row], s_inv[row], lr, al, ds, gus, guf);
                        const uint64_t node = (uint64_t)ix.node[row];
                        st4_wt(rs_dzb, (node * 2 * H + 4 * lr) * 4, dzs);
                        st4_wt(rs_dzb, (node * 2 * H + H + 4 * lr) * 4, dzf);
                        const int deg = sp.y - sp.x;
                        if (lr == 0) {
#pragma unroll
                            for (int k = 0; k < kInRegs; ++k)
                                if (k < deg) { st1_wt(a.alpha + sp.x + k, al[k]); st1_wt(a.dsc + sp.x + k, ds[k]); }
                        }
                    }
                }
                // region R has been free since the passes: the next tile's consumer lists
                os.c[tid] = n_c; os.sl[tid] = n_sl; os.gc[tid] = (uint8_t)n_gc;
            }
            STAMP(6);
            lds_barrier();
            cur = nxt; cur_start = nxt_start; cur_count = nxt_count; pp ^= 1;
            if (nxt.valid(pa)) {
                nxt.advance(pa, g, j, w);
                if (nxt.valid(pa)) { nxt_start = a.tile_start[nxt.t]; nxt_count = a.tile_count[nxt.t]; }
            }
        }
        STAMP(7);
        if (lvl > 1) {
            if (!grid_barrier<false>(pa.bar, gb, s_flag)) { if (threadIdx.x == 0) gb_store(pa.sticky, 4u); return; }
        }
        STAMP(8);
    }
    STAMP_FLUSH(a);

    // ---- the workgroup's small gradient partials (the slab row's tail: 2H d attn_u | 3H dbvc | 3H dbih | 3H dbhh)
    float* slab = pa.wg_slab + (int64_t)blockIdx.x * (6 * H * H + 11 * H);
    float* s_bias = reinterpret_cast<float*>(smem_raw + P::o_zhi);        // [2][9H]
    float* s_gu = s_bias + 2 * 9 * H;                                     // [8][2H]
    lds_barrier();
    {
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
        if constexpr (!WGW) {
            const int wi = wv & 3, jh = wv >> 2;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        slab[(p * H + wi * 16 + q * 4 + e) * 2 * H + (jh * 4 + jj) * 16 + r] = wacc[p][jj][e];
        }
        // bias sums: the four row quads of a lane's column meet by shuffles, the two row-tile waves of a column tile through LDS
        const int wc = wv % S::WPC, wr = wv / S::WPC, col = wc * 16 + r;
        float vals[9] = {v_r, v_z, v_n, b_r, b_z, b_n, b_r, b_z, h_n};        // dbvc (r z n) | dbih (r z n) | dbhh (r z n.r)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float v = vals[k];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (q == 0) s_bias[wr * 9 * H + k * H + col] = v;
        }
#pragma unroll
        for (int mk = LPR; mk < 64; mk <<= 1) {
            gus.x += __shfl_xor(gus.x, mk, 64); gus.y += __shfl_xor(gus.y, mk, 64); gus.z += __shfl_xor(gus.z, mk, 64); gus.w += __shfl_xor(gus.w, mk, 64);
            guf.x += __shfl_xor(guf.x, mk, 64); guf.y += __shfl_xor(guf.y, mk, 64); guf.z += __shfl_xor(guf.z, mk, 64); guf.w += __shfl_xor(guf.w, mk, 64);
        }
        if (lane < LPR) {
            st4(s_gu + wv * 2 * H + 4 * lane, gus);
            st4(s_gu + wv * 2 * H + H + 4 * lane, guf);
        }
    }
    lds_barrier();
    for (int i = threadIdx.x; i < 11 * H; i += kLT) {
        float v;
        if (i < 2 * H) {
            v = 0.f;
#pragma unroll
            for (int ww = 0; ww < kLW; ++ww) v += s_gu[ww * 2 * H + i];
        } else {
            v = s_bias[i - 2 * H] + s_bias[9 * H + i - 2 * H];
        }
        slab[6 * H * H + i] = v;
    }
}

// dWvc[g], d_attn_u[g], dbvc[g], dbih[g], dbhh[g] += the slab rows of slot g's workgroups, in workgroup order (four interleaved
// partial sums combined in a fixed order): deterministic
template <int H>
__global__ __launch_bounds__(256) void k_persist_reduce(PersistArgs pa) {
    constexpr int W = 6 * H * H + 11 * H;
    const int g = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W) return;
    const int b0 = pa.wg_begin[g], b1 = pa.wg_begin[g + 1];
    if (b1 <= b0) return;
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    const float* src = pa.wg_slab + i;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] += src[(int64_t)(b + k) * W];
    }
    for (int k = 0; b < b1; ++b, ++k) p[k] += src[(int64_t)b * W];
    const LevelX3Args& a = pa.a;
    const int s = i - 6 * H * H;
    float* dst = i < 6 * H * H ? a.dWvc + (int64_t)g * 6 * H * H + i
               : s < 2 * H ? a.d_attn_u + (int64_t)g * 2 * H + s
               : s < 5 * H ? a.dbvc + (int64_t)g * 3 * H + (s - 2 * H)
               : s < 8 * H ? a.dbih + (int64_t)g * 3 * H + (s - 5 * H) : a.dbhh + (int64_t)g * 3 * H + (s - 8 * H);
    *dst += (p[0] + p[1]) + (p[2] + p[3]);
}

}  // namespace mgv

#ifdef MGV_STAMPS
static unsigned long long* g_persist_stamps = nullptr;
extern "C" int mgv_diag_set_persist_stamps(void* p) { g_persist_stamps = static_cast<unsigned long long*>(p); return 0; }
#define MGV_SET_PERSIST_STAMPS(a) (a).stamps = g_persist_stamps
#else
#define MGV_SET_PERSIST_STAMPS(a)
#endif

namespace {

int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = -1;
    }
    return n;
}

// wg_begin: monotone, starts at 0, a slot with tiles has at least one workgroup, the grid fits the CUs (co-residency: one
// workgroup per CU by LDS size)
int check_roles(int T, const int32_t* wg_begin_host, const int32_t* slot_tile_ptr_host) {
    if (!wg_begin_host || !slot_tile_ptr_host || wg_begin_host[0] != 0) return MGV_EINVAL;
    for (int g = 0; g < T; ++g) {
        if (wg_begin_host[g + 1] < wg_begin_host[g]) return MGV_EINVAL;
        if (slot_tile_ptr_host[g + 1] > slot_tile_ptr_host[g] && wg_begin_host[g + 1] == wg_begin_host[g]) return MGV_EINVAL;
    }
    const int cus = cu_count();
    if (cus <= 0 || wg_begin_host[T] < 1 || wg_begin_host[T] > cus) return MGV_EUNSUPPORTED;
    return MGV_OK;
}

}  // namespace

extern "C" int mgv_sweep_persist_sync_bytes(void) { return (int)sizeof(mgv::GridBarState); }

extern "C" int mgv_sweep_persist_max_grid(void) { return cu_count(); }

extern "C" int mgv_func_sweep_fwd_persist_x3(int H, int64_t N, int T, int num_levels, const int32_t* key_tile_ptr,
                                             const int32_t* wg_begin_host, const int32_t* slot_tile_ptr_host, const int32_t* order,
                                             const int32_t* order_span, const int32_t* tile_start, const int32_t* tile_count,
                                             const int32_t* in_ptr, const int32_t* in_src, const float* hs, float* hf,
                                             const float* attn_u, const void* wpack_bf16, const float* bvc, const float* bih,
                                             const float* bhh, void* sync_ws, void* sticky_status, void* stream) {
    MGV_CHECK_ARG(N >= 0 && T >= 1 && T <= mgv::kMaxSlots && num_levels >= 0 && hs && hf && attn_u && wpack_bf16 && bvc && bih && bhh && in_ptr);
    if (H != 64) return MGV_EUNSUPPORTED;
    if ((uint64_t)N * H * 4 >= (1ull << 32)) return MGV_EUNSUPPORTED;          // buffer descriptors address 4 GB
    if (num_levels <= 1 || N == 0) return MGV_OK;
    MGV_CHECK_ARG(key_tile_ptr && order && order_span && tile_start && tile_count && in_src && sync_ws && sticky_status);
    const int rc = check_roles(T, wg_begin_host, slot_tile_ptr_host);
    if (rc != MGV_OK) return rc;
    mgv::PersistArgs pa{};
    mgv::LevelX3Args& a = pa.a;
    MGV_SET_PERSIST_STAMPS(a);
    a.N = N; a.T = T; a.order = order; a.order_span = order_span; a.span_ints = 4; a.tile_start = tile_start; a.tile_count = tile_count;
    a.in_ptr = in_ptr; a.in_src = in_src; a.hs = hs; a.hf = hf; a.attn_u = attn_u; a.wpack = static_cast<const __bf16*>(wpack_bf16);
    a.bvc = bvc; a.bih = bih; a.bhh = bhh;
    pa.num_levels = num_levels; pa.key_tile_ptr = key_tile_ptr;
    for (int g = 0; g <= mgv::kMaxSlots; ++g) pa.wg_begin[g] = wg_begin_host[g < T ? g : T];
    pa.bar = static_cast<mgv::GridBarState*>(sync_ws); pa.sticky = static_cast<unsigned*>(sticky_status);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const hipError_t e = hipMemsetAsync(sync_ws, 0, sizeof(mgv::GridBarState), st);
    if (e != hipSuccess) return (int)e;
    using P = mgv::PFwdSmem<64>;
    static bool set = false;
    if (!set) { hipFuncSetAttribute(reinterpret_cast<const void*>(mgv::k_sweep_fwd_persist<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    hipLaunchKernelGGL(mgv::k_sweep_fwd_persist<64>, dim3(wg_begin_host[T]), dim3(mgv::kLT), P::bytes, st, pa);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_sweep_persist_slab_floats(int H, int grid) { return grid * (6 * H * H + 11 * H); }

extern "C" int mgv_func_sweep_bwd_persist_x3(int H, int64_t N, int T, int num_levels, const int32_t* key_tile_ptr,
                                             const int32_t* wg_begin_host, const int32_t* slot_tile_ptr_host, const int32_t* order,
                                             const int32_t* order_span, const int32_t* tile_start, const int32_t* tile_count,
                                             const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                             const int32_t* out_dst, const int32_t* out_slot, const uint8_t* gslot, const float* hs,
                                             const float* hf, const float* attn_u, const void* wpack_bf16, const float* bvc,
                                             const float* bih, const float* bhh, const float* ghf, float* ghs, float* dzb,
                                             float* alpha, float* dsc, float* d_attn_u, float* dWvc, float* dbvc, float* dbih,
                                             float* dbhh, float* wg_slab, int64_t wg_slab_floats, int skip_inactive_longer_than,
                                             void* sync_ws, void* sticky_status, void* stream) {
    MGV_CHECK_ARG(N >= 0 && T >= 1 && T <= mgv::kMaxSlots && num_levels >= 0 && hs && hf && attn_u && wpack_bf16 && bvc && bih && bhh);
    MGV_CHECK_ARG(in_ptr && out_ptr && gslot && ghf && ghs && dzb && d_attn_u && dWvc && dbvc && dbih && dbhh);
    if (H != 64) return MGV_EUNSUPPORTED;
    if ((uint64_t)N * 2 * H * 4 >= (1ull << 32)) return MGV_EUNSUPPORTED;      // buffer descriptors address 4 GB
    if (N == 0) return MGV_OK;
    mgv::PersistArgs pa{};
    mgv::LevelX3Args& a = pa.a;
    MGV_SET_PERSIST_STAMPS(a);
    a.N = N; a.T = T; a.order = order; a.order_span = order_span; a.span_ints = 4; a.tile_start = tile_start; a.tile_count = tile_count;
    a.in_ptr = in_ptr; a.in_src = in_src; a.hs = hs; a.hf = const_cast<float*>(hf); a.attn_u = attn_u;
    a.wpack = static_cast<const __bf16*>(wpack_bf16); a.bvc = bvc; a.bih = bih; a.bhh = bhh;
    a.out_ptr = out_ptr; a.out_dst = out_dst; a.out_slot = out_slot; a.gslot = gslot;
    a.ghf = ghf; a.ghs = ghs; a.dzb = dzb; a.alpha = alpha; a.dsc = dsc; a.d_attn_u = d_attn_u; a.dWvc = dWvc; a.dbvc = dbvc;
    a.dbih = dbih; a.dbhh = dbhh; a.skip_inactive = skip_inactive_longer_than;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (num_levels > 1) {
        MGV_CHECK_ARG(key_tile_ptr && order && order_span && tile_start && tile_count && in_src && out_dst && out_slot && alpha && dsc &&
                      sync_ws && sticky_status && wg_slab);
        const int rc = check_roles(T, wg_begin_host, slot_tile_ptr_host);
        if (rc != MGV_OK) return rc;
        const int grid = wg_begin_host[T];
        MGV_CHECK_ARG(wg_slab_floats >= (int64_t)grid * (6 * H * H + 11 * H));
        pa.num_levels = num_levels; pa.key_tile_ptr = key_tile_ptr;
        for (int g = 0; g <= mgv::kMaxSlots; ++g) pa.wg_begin[g] = wg_begin_host[g < T ? g : T];
        pa.bar = static_cast<mgv::GridBarState*>(sync_ws); pa.sticky = static_cast<unsigned*>(sticky_status); pa.wg_slab = wg_slab;
        const hipError_t e = hipMemsetAsync(sync_ws, 0, sizeof(mgv::GridBarState), st);
        if (e != hipSuccess) return (int)e;
        using P = mgv::PBwdSmem<64>;
        static int wgw = -1;             // MGV_PERSIST_WGW=1: the 12-wave variant (four weight-gradient waves); measured slower
        if (wgw < 0) {
            const char* e_ = getenv("MGV_PERSIST_WGW");
            wgw = (e_ && e_[0] == '1') ? 1 : 0;
            hipFuncSetAttribute(reinterpret_cast<const void*>(mgv::k_sweep_bwd_persist<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(mgv::k_sweep_bwd_persist<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        if (wgw) hipLaunchKernelGGL((mgv::k_sweep_bwd_persist<64, true>), dim3(grid), dim3(mgv::kPBT), P::bytes, st, pa);
        else hipLaunchKernelGGL((mgv::k_sweep_bwd_persist<64, false>), dim3(grid), dim3(mgv::kLT), P::bytes, st, pa);
        constexpr int W = 6 * 64 * 64 + 11 * 64;
        hipLaunchKernelGGL(mgv::k_persist_reduce<64>, dim3((W + 255) / 256, T), dim3(256), 0, st, pa);
    }
    // nodes the sweep never updates: their hs rows' pull (as in mgv_func_sweep_bwd_x3)
    const int rows_per_block = mgv::kThreads / (H / 4);
    const int grid_i = mgv::grid_for((N + rows_per_block - 1) / rows_per_block, 8);
    hipLaunchKernelGGL(mgv::k_level_pull_inactive_x3<64>, dim3(grid_i), dim3(mgv::kThreads), 0, st, a);
    MGV_LAUNCH_RET();
}

// the sticky status word of the persistent sweeps, read synchronously: 0 = every launch so far completed its barriers
extern "C" int mgv_sweep_persist_status(const void* sticky_status, void* stream) {
    MGV_CHECK_ARG(sticky_status);
    unsigned v = 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(&v, sticky_status, sizeof(v), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
    return v == 0 ? MGV_OK : 1000 + (int)v;
}
