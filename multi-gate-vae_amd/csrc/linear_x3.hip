// Linear layers over node rows on bf16x3 split-precision MFMA (mgv_x3.h): the hs_linear / hs_decompose / VAE
// heads / readout layers of dg_ae_model_aig.py:50-56,100-106 and their input- and weight-gradients.  The fp32
// kernels of dense.hip price these streaming layers at the fp32-MFMA rate (K=128, M=64: 69 GFLOP per pass);
// here they are HBM-bound: rows are split into bf16 hi/lo planes in LDS once, the forward keeps its weight
// fragments in registers for the whole launch, the weight gradient reads both operands transposed from the
// row-major planes, and the next tile's rows are in flight (registers) while the current tile is multiplied.
#include "mgv_x3.h"
#include "mgv_slab.h"
#include "../../include/mgvae_hip.h"

namespace mgv {

struct LinX3Args {
    int64_t N;
    const float* X1; int K1; int ld1;
    const float* X2; int K2; int ld2;
    const __bf16* wpack;    // forward: [2][M*K] = W_hi, W_lo in fragment order (blocks (row tile, k-step))
    const float* b;
    float* Y; int ldy;
    const float* R; int ldr;     // optional residual rows added to Y on the way out
    const float* dY; int lddy;
    float* dW; float* db;
    float* slab;            // weight gradient: [gridDim][M*K + M] per-workgroup partials (dW row-major, then db)
    // grouped mode (order != nullptr): the rows are the level sweep's tiles — row r of tile t is NODE order[tile_start[t] + r]
    // (r < tile_count[t]); X / Y / R / dY rows are indexed by node; the forward takes tile t's weights from the pack of slot
    // tile_slot[t] (g_wstride elements apart) and its bias from b + tile_slot[t] * M.  tile_list (nullable) names the tiles.
    const int32_t* order; const int32_t* tile_start; const int32_t* tile_count; const int32_t* tile_slot; const int32_t* tile_list;
    int64_t g_ntiles; int64_t g_wstride;
};

// the wave split of a 64-row tile over M output columns (mgv_common.h WaveSplit without its H <= 128 bound: M = 3H = 192 here)
template <int M>
struct LinSplit {
    static_assert(M % 16 == 0, "column tiles of 16");
    static constexpr int HC = M / 16;
    static constexpr int WPC = HC < 4 ? HC : 4;
    static constexpr int WPR = 4 / WPC;
    static constexpr int RTW = 4 / WPR;
    static constexpr int HCW = HC / WPC;
    static_assert(HCW * WPC == HC, "column tiles must split over the waves");
};

__device__ __forceinline__ float4 f4x(const f32x4& v) { return make_float4(v[0], v[1], v[2], v[3]); }

// ---------------------------------------------------------------------------------------------- forward
template <int M, int K>
struct LinFwdGeom {
    using S = LinSplit<M>;                                    // 4 waves
    static constexpr int KS = K / 32;
    static constexpr int LDP = K + 8;                         // bf16 plane row
    static constexpr int PB = kTileRows * LDP * 2;
    static constexpr int LDY = M + 4;
    static constexpr int o_y = 2 * PB;
    static constexpr int smem_bytes = o_y + kTileRows * LDY * 4;
    static constexpr int F4 = kTileRows * K / 4;              // float4 per tile
    static constexpr int PF = F4 / kThreads;
    static_assert(PF * kThreads == F4 && K % 32 == 0, "tile must split over the threads");
};

template <int M, int K, bool GROUPED = false>
__global__ __launch_bounds__(kThreads) void k_linear_fwd_x3(LinX3Args a) {
    using G = LinFwdGeom<M, K>;
    using S = typename G::S;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* x_hi = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* x_lo = reinterpret_cast<__bf16*>(smem_raw + G::PB);
    float* s_y = reinterpret_cast<float*>(smem_raw + G::o_y);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
    // this wave's weight fragments: column tiles wc*HCW.., all k-steps, hi and lo (grouped mode: reloaded per tile, the tile's slot)
    bf16x8 wh[S::HCW][G::KS], wl[S::HCW][G::KS];
    float bias[S::HCW];
    constexpr bool grouped = GROUPED;          // (its own instantiation: the plain kernels carry none of this)
    auto load_weights = [&](const __bf16* wp, const float* bp) {
#pragma unroll
        for (int j = 0; j < S::HCW; ++j) {
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const int wo = (((wc * S::HCW + j) * G::KS) + ks) * 512 + lane * 8;
                wh[j][ks] = ldfrag(wp + wo); wl[j][ks] = ldfrag(wp + M * K + wo);
            }
            bias[j] = bp ? bp[(wc * S::HCW + j) * 16 + r] : 0.f;
        }
    };
    if (!grouped) load_weights(a.wpack, a.b);
    const int64_t ntiles = grouped ? a.g_ntiles : (a.N + kTileRows - 1) / kTileRows;
    // rows of a tile -> node ids: plain tiles are 64 consecutive nodes; grouped tiles go through the sweep's order list
    auto tile_span = [&](int64_t tile, int64_t& first, int& cnt, int& slot) {
        if (grouped) {
            const int64_t t = a.tile_list ? (int64_t)a.tile_list[tile] : tile;
            first = a.tile_start[t]; cnt = a.tile_count[t]; slot = a.tile_slot ? a.tile_slot[t] : 0;
        } else {
            first = tile * kTileRows; cnt = (int)(a.N - first < kTileRows ? a.N - first : kTileRows); slot = 0;
        }
    };
    auto node_of = [&](int64_t first, int cnt, int row) -> int64_t { return row < cnt ? (grouped ? (int64_t)a.order[first + row] : first + row) : -1; };
    f32x4 pf[G::PF];
    auto prefetch = [&](int64_t tile) {
        int64_t first; int cnt, slot;
        tile_span(tile, first, cnt, slot);
#pragma unroll
        for (int u = 0; u < G::PF; ++u) {
            const int f = tid + u * kThreads;
            const int row = f / (K / 4), c4 = (f % (K / 4)) * 4;
            const int64_t node = node_of(first, cnt, row);
            if (node >= 0) pf[u] = *reinterpret_cast<const f32x4*>(c4 < a.K1 ? a.X1 + node * a.ld1 + c4 : a.X2 + node * a.ld2 + (c4 - a.K1));
            else pf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles) prefetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        int64_t first; int cnt, slot;
        tile_span(tile, first, cnt, slot);
        if (grouped) load_weights(a.wpack + (int64_t)slot * a.g_wstride, a.b ? a.b + (int64_t)slot * M : nullptr);
#pragma unroll
        for (int u = 0; u < G::PF; ++u) {
            const int f = tid + u * kThreads;
            const int row = f / (K / 4), c4 = (f % (K / 4)) * 4;
            bf16x4 hi, lo;
            split4(f4x(pf[u]), hi, lo);
            st_bf4(x_hi + row * G::LDP + c4, hi); st_bf4(x_lo + row * G::LDP + c4, lo);
        }
        lds_barrier();
        if (tile + gridDim.x < ntiles) prefetch(tile + gridDim.x);
        f32x4 acc[S::RTW][S::HCW];
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int j = 0; j < S::HCW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
            for (int i = 0; i < S::RTW; ++i) {
                const int off = ((wr * S::RTW + i) * 16 + r) * G::LDP + 32 * ks + 8 * q;
                const bf16x8 xh = ldfrag(x_hi + off), xl = ldfrag(x_lo + off);
#pragma unroll
                for (int j = 0; j < S::HCW; ++j) mma_x3(acc[i][j], xh, xl, wh[j][ks], wl[j][ks]);
            }
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int j = 0; j < S::HCW; ++j) {
                const int col = (wc * S::HCW + j) * 16 + r;
#pragma unroll
                for (int e = 0; e < 4; ++e) s_y[((wr * S::RTW + i) * 16 + q * 4 + e) * G::LDY + col] = acc[i][j][e] + bias[j];
            }
        lds_barrier();
        for (int i = tid; i < kTileRows * (M / 4); i += kThreads) {
            const int row = i / (M / 4), c4 = (i % (M / 4)) * 4;
            const int64_t node = node_of(first, cnt, row);
            if (node >= 0) {
                float4 v = ld4(s_y + row * G::LDY + c4);
                if (a.R) v = add4(v, ld4(a.R + node * a.ldr + c4));
                st4(a.Y + node * a.ldy + c4, v);
            }
        }
        // the planes are rewritten only after this barrier pair; s_y only after the next tile's first barrier
    }
}

// ---------------------------------------------------------------------------------------------- weight gradient
// dW[M][K] += dY^T [X1|X2], db[M] += colsum(dY).  NW waves in a WI x WJ grid over the (M/16) x (K/16) output tiles.
template <int M, int K, int NW, int WI>
struct LinWgGeom {
    static constexpr int NT = 64 * NW;
    static constexpr int TI = M / 16, TJ = K / 16, WJ = NW / WI;
    static constexpr int ITW = TI / WI, JTW = TJ / WJ;
    static_assert(ITW * WI == TI && JTW * WJ == TJ && ITW >= 1 && JTW >= 1, "wave grid must tile the output");
    static constexpr int LDG = M + 8, LDX = K + 8;
    static constexpr int GPB = kTileRows * LDG * 2, XPB = kTileRows * LDX * 2;
    static constexpr int F4G = kTileRows * M / 4, F4X = kTileRows * K / 4;
    static constexpr int PFG = (F4G + NT - 1) / NT, PFX = (F4X + NT - 1) / NT;
    static_assert(NT % (M / 4) == 0, "a thread must keep its dY column across its loads");
    static constexpr int o_db = 2 * GPB + 2 * XPB;
    static constexpr int smem_bytes = o_db + M * 4;
};

template <int M, int K, int NW, int WI, bool GROUPED = false>
__global__ __launch_bounds__(64 * NW) void k_linear_wgrad_x3(LinX3Args a) {
    using G = LinWgGeom<M, K, NW, WI>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* g_hi = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* g_lo = reinterpret_cast<__bf16*>(smem_raw + G::GPB);
    __bf16* x_hi = reinterpret_cast<__bf16*>(smem_raw + 2 * G::GPB);
    __bf16* x_lo = reinterpret_cast<__bf16*>(smem_raw + 2 * G::GPB + G::XPB);
    float* s_db = reinterpret_cast<float*>(smem_raw + G::o_db);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int it0 = (w / G::WJ) * G::ITW, jt0 = (w % G::WJ) * G::JTW;
    for (int i = tid; i < M; i += G::NT) s_db[i] = 0.f;
    f32x4 acc[G::ITW][G::JTW];
#pragma unroll
    for (int i = 0; i < G::ITW; ++i)
#pragma unroll
        for (int j = 0; j < G::JTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 dbs = zero4();            // column sums of dY over this thread's rows (its column quad is fixed)
    f32x4 pg[G::PFG], px[G::PFX];
    constexpr bool grouped = GROUPED;
    const int64_t ntiles = grouped ? a.g_ntiles : (a.N + kTileRows - 1) / kTileRows;
    auto prefetch = [&](int64_t tile) {
        int64_t first = tile * kTileRows;
        int cnt = (int)(a.N - first < kTileRows ? a.N - first : kTileRows);
        if (grouped) {
            const int64_t t = a.tile_list ? (int64_t)a.tile_list[tile] : tile;
            first = a.tile_start[t]; cnt = a.tile_count[t];
        }
#pragma unroll
        for (int u = 0; u < G::PFG; ++u) {
            const int f = tid + u * G::NT;
            const int row = f / (M / 4), c4 = (f % (M / 4)) * 4;
            const int64_t node = row < cnt ? (grouped ? (int64_t)a.order[first + row] : first + row) : -1;
            if (f < G::F4G && node >= 0) pg[u] = *reinterpret_cast<const f32x4*>(a.dY + node * a.lddy + c4);
            else pg[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < G::PFX; ++u) {
            const int f = tid + u * G::NT;
            const int row = f / (K / 4), c4 = (f % (K / 4)) * 4;
            const int64_t node = row < cnt ? (grouped ? (int64_t)a.order[first + row] : first + row) : -1;
            if (f < G::F4X && node >= 0) px[u] = *reinterpret_cast<const f32x4*>(c4 < a.K1 ? a.X1 + node * a.ld1 + c4 : a.X2 + node * a.ld2 + (c4 - a.K1));
            else px[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles) prefetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int u = 0; u < G::PFG; ++u) {
            const int f = tid + u * G::NT;
            if (f < G::F4G) {
                const int row = f / (M / 4), c4 = (f % (M / 4)) * 4;
                const float4 v = f4x(pg[u]);
                dbs = add4(dbs, v);
                bf16x4 hi, lo;
                split4(v, hi, lo);
                st_bf4(g_hi + row * G::LDG + c4, hi); st_bf4(g_lo + row * G::LDG + c4, lo);
            }
        }
#pragma unroll
        for (int u = 0; u < G::PFX; ++u) {
            const int f = tid + u * G::NT;
            if (f < G::F4X) {
                const int row = f / (K / 4), c4 = (f % (K / 4)) * 4;
                bf16x4 hi, lo;
                split4(f4x(px[u]), hi, lo);
                st_bf4(x_hi + row * G::LDX + c4, hi); st_bf4(x_lo + row * G::LDX + c4, lo);
            }
        }
        lds_barrier();
        if (tile + gridDim.x < ntiles) prefetch(tile + gridDim.x);
#pragma unroll
        for (int ks = 0; ks < kTileRows / 32; ++ks) {
            bf16x8 bh[G::JTW], bl[G::JTW];
#pragma unroll
            for (int j = 0; j < G::JTW; ++j) {
                bh[j] = ldfrag_tr2(x_hi, G::LDX, 32 * ks, (jt0 + j) * 16);
                bl[j] = ldfrag_tr2(x_lo, G::LDX, 32 * ks, (jt0 + j) * 16);
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
    // per-workgroup partials to this workgroup's slab row (plain stores); k_slab_sum adds the rows in a fixed order
    float* slab = a.slab + (int64_t)blockIdx.x * (M * K + M);
#pragma unroll
    for (int i = 0; i < G::ITW; ++i)
#pragma unroll
        for (int j = 0; j < G::JTW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                slab[((it0 + i) * 16 + q * 4 + e) * K + (jt0 + j) * 16 + r] = acc[i][j][e];
    if (a.db) {
        // column sums of dY: one float4 per thread into the (dead) planes, then a fixed-order sum over the threads of a column quad
        float4* s_part = reinterpret_cast<float4*>(smem_raw);
        if (tid < G::F4G) s_part[tid] = dbs;
        __syncthreads();
        constexpr int QPR = M / 4;                       // column quads per row; thread t holds quad t % QPR
        if (tid < M) {
            const int c4 = tid / 4, e = tid % 4;
            float sum = 0.f;
            for (int t = c4; t < G::F4G && t < G::NT; t += QPR) { const float4 v = s_part[t]; sum += e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
            slab[M * K + tid] = sum;
        }
    }
}

// fp32 matrix -> bf16 hi/lo planes in MFMA fragment order: block (row tile, k-step) = 512 elements, lane 16q+r holds
// A[16 rt + r][32 ks + 8q .. +7] where A = W (R x K, leading dimension ldw) or its transpose view A[i][k] = W[k][i]
__global__ __launch_bounds__(256) void k_wpack_bf16x3(const float* W, int R, int K, int ldw, int transpose, __bf16* hi, __bf16* lo) {
    const int total = R * K, ksn = K / 32;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
        const int blk = o >> 9, within = o & 511, lane = within >> 3, e = within & 7;
        const int rt = blk / ksn, ks = blk % ksn;
        const int row = rt * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + e;
        const float v = transpose ? W[(int64_t)k * ldw + row] : W[(int64_t)row * ldw + k];
        __bf16 h, l;
        split_bf16(v, h, l);
        hi[o] = h; lo[o] = l;
    }
}

template <int M, int K, bool GROUPED = false>
int launch_linear_fwd_x3(const LinX3Args& a, hipStream_t st) {
    using G = LinFwdGeom<M, K>;
    static bool set = false;
    if (!set) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_fwd_x3<M, K, GROUPED>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    const int64_t ntiles = a.order ? a.g_ntiles : (a.N + kTileRows - 1) / kTileRows;
    int per_cu = 160 * 1024 / G::smem_bytes;
    per_cu = per_cu > 4 ? 4 : per_cu;
    hipLaunchKernelGGL((k_linear_fwd_x3<M, K, GROUPED>), dim3(grid_for(ntiles, per_cu)), dim3(kThreads), G::smem_bytes, st, a);
    MGV_LAUNCH_RET();
}

template <int M, int K, int NW, int WI, bool GROUPED = false>
int launch_linear_wgrad_x3(const LinX3Args& a, int64_t ws_floats, hipStream_t st) {
    using G = LinWgGeom<M, K, NW, WI>;
    static bool set = false;
    if (!set) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_wgrad_x3<M, K, NW, WI, GROUPED>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    const int64_t ntiles = a.order ? a.g_ntiles : (a.N + kTileRows - 1) / kTileRows;
    int per_cu = 160 * 1024 / G::smem_bytes;
    const int cap = 1024 / G::NT;                    // 16 waves per CU
    per_cu = per_cu > cap ? cap : per_cu;
    const int grid = grid_for(ntiles, per_cu);
    if (a.slab == nullptr || ws_floats < (int64_t)grid * (M * K + M)) return MGV_EINVAL;
    hipLaunchKernelGGL((k_linear_wgrad_x3<M, K, NW, WI, GROUPED>), dim3(grid), dim3(G::NT), G::smem_bytes, st, a);
    launch_slab_sum<float, float>(a.slab, grid, M * K + M, M * K, a.dW, st);
    if (a.db) launch_slab_sum<float, float>(a.slab + M * K, grid, M * K + M, M, a.db, st);
    MGV_LAUNCH_RET();
}

}  // namespace mgv

extern "C" int mgv_linear_x3_supported(int M, int K) {
    return (M == 64 && K == 128) || (M == 128 && K == 64) || (M == 64 && K == 64) || (M == 64 && K == 32) || (M == 32 && K == 64) ||
           (M == 32 && K == 32);
}

static int linear_fwd_x3_impl(int64_t N, const float* X1, int K1, int ld1, const float* X2, int K2, int ld2, const void* wpack_bf16,
                              const float* b, int M, const float* R, int ldr, float* Y, int ldy, void* stream) {
    MGV_CHECK_ARG(N >= 0 && X1 && wpack_bf16 && Y && K1 > 0 && K2 >= 0 && (K2 == 0 || X2));
    MGV_CHECK_ARG(K1 % 4 == 0 && K2 % 4 == 0 && ld1 >= K1 && ld1 % 4 == 0 && (K2 == 0 || (ld2 >= K2 && ld2 % 4 == 0)) && ldy >= M && ldy % 4 == 0);
    MGV_CHECK_ARG(R == nullptr || (ldr >= M && ldr % 4 == 0));
    if (N == 0) return MGV_OK;
    mgv::LinX3Args a{};
    a.N = N; a.X1 = X1; a.K1 = K1; a.ld1 = ld1; a.X2 = X2; a.K2 = K2; a.ld2 = ld2; a.wpack = static_cast<const __bf16*>(wpack_bf16);
    a.b = b; a.Y = Y; a.ldy = ldy; a.R = R; a.ldr = ldr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int K = K1 + K2;
#define MGV_LF(MM, KK) if (M == MM && K == KK) return mgv::launch_linear_fwd_x3<MM, KK>(a, st);
    MGV_LF(64, 128) MGV_LF(128, 64) MGV_LF(64, 64) MGV_LF(64, 32) MGV_LF(32, 64) MGV_LF(32, 32)
#undef MGV_LF
    return MGV_EUNSUPPORTED;
}

extern "C" int mgv_linear_fwd_x3(int64_t N, const float* X1, int K1, int ld1, const float* X2, int K2, int ld2,
                                 const void* wpack_bf16, const float* b, int M, float* Y, int ldy, void* stream) {
    return linear_fwd_x3_impl(N, X1, K1, ld1, X2, K2, ld2, wpack_bf16, b, M, nullptr, 0, Y, ldy, stream);
}

extern "C" int mgv_linear_fwd_x3_res(int64_t N, const float* X1, int K1, int ld1, const float* X2, int K2, int ld2,
                                     const void* wpack_bf16, const float* b, int M, const float* R, int ldr, float* Y, int ldy,
                                     void* stream) {
    MGV_CHECK_ARG(R != nullptr);
    return linear_fwd_x3_impl(N, X1, K1, ld1, X2, K2, ld2, wpack_bf16, b, M, R, ldr, Y, ldy, stream);
}

// floats of workspace mgv_linear_wgrad_x3 needs: one row of M*K + M partials per workgroup (at most 4 workgroups per CU)
extern "C" int mgv_linear_wgrad_x3_ws_floats(int M, int K, int64_t N) {
    if (M <= 0 || K <= 0 || N < 0) return 0;
    const int64_t ntiles = (N + mgv::kTileRows - 1) / mgv::kTileRows;
    return mgv::grid_for(ntiles, 4) * (M * K + M);
}

extern "C" int mgv_linear_wgrad_x3(int64_t N, const float* X1, int K1, int ld1, const float* X2, int K2, int ld2,
                                   const float* dY, int lddy, int M, float* dW, float* db, float* workspace,
                                   int64_t workspace_floats, void* stream) {
    MGV_CHECK_ARG(N >= 0 && X1 && dY && dW && K1 > 0 && K2 >= 0 && (K2 == 0 || X2));
    MGV_CHECK_ARG(K1 % 4 == 0 && K2 % 4 == 0 && ld1 % 4 == 0 && (K2 == 0 || ld2 % 4 == 0) && lddy % 4 == 0 && lddy >= M);
    if (N == 0) return MGV_OK;
    mgv::LinX3Args a{};
    a.N = N; a.X1 = X1; a.K1 = K1; a.ld1 = ld1; a.X2 = X2; a.K2 = K2; a.ld2 = ld2; a.dY = dY; a.lddy = lddy; a.dW = dW; a.db = db; a.slab = workspace;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int K = K1 + K2;
#define MGV_WGX(MM, KK, NW, WI) if (M == MM && K == KK) return mgv::launch_linear_wgrad_x3<MM, KK, NW, WI>(a, workspace_floats, st);
    MGV_WGX(64, 128, 8, 2) MGV_WGX(128, 64, 8, 4) MGV_WGX(64, 64, 8, 2) MGV_WGX(64, 32, 8, 4) MGV_WGX(32, 64, 8, 2) MGV_WGX(32, 32, 4, 2)
#undef MGV_WGX
    return MGV_EUNSUPPORTED;
}

extern "C" int mgv_wpack_bf16x3(const float* W, int R, int K, int ldw, int transpose, void* hi, void* lo, void* stream) {
    MGV_CHECK_ARG(W && hi && lo && R > 0 && K > 0 && R % 16 == 0 && K % 32 == 0 && ldw >= (transpose ? R : K));
    const int total = R * K;
    hipLaunchKernelGGL(mgv::k_wpack_bf16x3, dim3((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), W, R, K, ldw, transpose, static_cast<__bf16*>(hi), static_cast<__bf16*>(lo));
    MGV_LAUNCH_RET();
}

/* ---- grouped Linear over the level sweep's tiles (num_rounds > 1: gh = W_hh[slot] h_prev + b_hh[slot] per updated gate with its OWN
 * aggregator's GRU weights, dg_ae_model_aig.py:88-94; one launch instead of an index_select / three Linears / index_copy per gate type).
 * H = 64 shapes: (M, K) = (192, 64) forward, (64, 192) input gradient, (192, 64) weight gradient of ONE slot's tile list. */
extern "C" int mgv_grouped_linear_supported(int M, int K) { return (M == 192 && K == 64) || (M == 64 && K == 192); }

extern "C" int mgv_grouped_linear_fwd_x3(int64_t ntiles, const int32_t* tile_list, const int32_t* order, const int32_t* tile_start,
                                         const int32_t* tile_count, const int32_t* tile_slot, const float* X, int K, int ldx,
                                         const void* wpack_bf16, const float* b, int M, const float* R, int ldr, float* Y, int ldy, void* stream) {
    MGV_CHECK_ARG(ntiles >= 0 && order && tile_start && tile_count && tile_slot && X && wpack_bf16 && Y);
    MGV_CHECK_ARG(K % 4 == 0 && ldx >= K && ldx % 4 == 0 && ldy >= M && ldy % 4 == 0 && (R == nullptr || (ldr >= M && ldr % 4 == 0)));
    if (ntiles == 0) return MGV_OK;
    mgv::LinX3Args a{};
    a.N = 0; a.X1 = X; a.K1 = K; a.ld1 = ldx; a.wpack = static_cast<const __bf16*>(wpack_bf16); a.b = b; a.Y = Y; a.ldy = ldy; a.R = R; a.ldr = ldr;
    a.order = order; a.tile_start = tile_start; a.tile_count = tile_count; a.tile_slot = tile_slot; a.tile_list = tile_list;
    a.g_ntiles = ntiles; a.g_wstride = (int64_t)2 * M * K;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (M == 192 && K == 64) return mgv::launch_linear_fwd_x3<192, 64, true>(a, st);
    if (M == 64 && K == 192) return mgv::launch_linear_fwd_x3<64, 192, true>(a, st);
    return MGV_EUNSUPPORTED;
}

extern "C" int mgv_grouped_linear_wgrad_x3_ws_floats(int M, int K, int64_t ntiles) {
    if (M <= 0 || K <= 0 || ntiles < 0) return 0;
    return mgv::grid_for(ntiles, 4) * (M * K + M);
}

extern "C" int mgv_grouped_linear_wgrad_x3(int64_t ntiles, const int32_t* tile_list, const int32_t* order, const int32_t* tile_start,
                                           const int32_t* tile_count, const float* X, int K, int ldx, const float* dY, int lddy, int M,
                                           float* dW, float* db, float* workspace, int64_t workspace_floats, void* stream) {
    MGV_CHECK_ARG(ntiles >= 0 && order && tile_start && tile_count && X && dY && dW && K % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddy >= M);
    if (ntiles == 0) return MGV_OK;
    mgv::LinX3Args a{};
    a.N = 0; a.X1 = X; a.K1 = K; a.ld1 = ldx; a.dY = dY; a.lddy = lddy; a.dW = dW; a.db = db; a.slab = workspace;
    a.order = order; a.tile_start = tile_start; a.tile_count = tile_count; a.tile_list = tile_list; a.g_ntiles = ntiles;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (M == 192 && K == 64) return mgv::launch_linear_wgrad_x3<192, 64, 6, 3, true>(a, workspace_floats, st);
    return MGV_EUNSUPPORTED;
}
