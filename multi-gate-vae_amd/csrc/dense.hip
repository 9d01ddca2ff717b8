// Row-streaming dense operators on [N, *] node matrices: Linear forward (also used for dgrad with the
// transposed weight), Linear weight/bias gradient, CSR gather-sum.  HBM-bound: every row is read once
// and written once; the small weight matrix is streamed from L2 as MFMA B fragments.
#include "mgv_common.h"
#include "mgv_slab.h"
#include "../../include/mgvae_hip.h"

namespace mgv {

struct LinArgs {
    int64_t N;
    const float* X1; int K1; int ld1;   // first K1 input columns (row stride ld1 floats)
    const float* X2; int K2; int ld2;   // next K2 input columns (may be null / 0): fused torch.cat
    const float* W;                     // [M][K1+K2]
    const float* b;                     // [M] or null
    float* Y; int ldy;                  // [N][M] with row stride ldy
    const float* dY; int lddy;          // wgrad: [N][M]
    float* dW; float* db;               // wgrad accumulators
};

// Y = [X1|X2] W^T + b,  M output columns (template), K = K1+K2 runtime (multiple of 16)
template <int M>
__global__ __launch_bounds__(kThreads) void k_linear_fwd(LinArgs a) {
    using S = WaveSplit<M>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = a.K1 + a.K2;
    const int LDX = K + 4;
    float* s_x = smem;
    float* s_y = smem + kTileRows * LDX;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wc = w % S::WPC, wr = w / S::WPC;
    const int kq = K / 4;                 // float4 per input row
    const int64_t ntiles = (a.N + kTileRows - 1) / kTileRows;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * kTileRows;
        for (int i = tid; i < kTileRows * kq; i += kThreads) {
            const int row = i / kq, c4 = (i % kq) * 4;
            const int64_t node = base + row;
            float4 v = zero4();
            if (node < a.N) v = c4 < a.K1 ? ld4(a.X1 + node * a.ld1 + c4) : ld4(a.X2 + node * a.ld2 + (c4 - a.K1));
            st4(s_x + row * LDX + c4, v);
        }
        __syncthreads();
        f32x4 acc[S::RTW][S::HCW];
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int j = 0; j < S::HCW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kb = 0; kb < K; kb += 16) {
            float4 xa[S::RTW];
#pragma unroll
            for (int i = 0; i < S::RTW; ++i) xa[i] = ld4(s_x + ((wr * S::RTW + i) * 16 + r) * LDX + kb + 4 * q);
#pragma unroll
            for (int j = 0; j < S::HCW; ++j) {
                const int col = (wc * S::HCW + j) * 16 + r;
                const float4 wv = ld4(a.W + (int64_t)col * K + kb + 4 * q);
#pragma unroll
                for (int i = 0; i < S::RTW; ++i) mma_kblock(acc[i][j], xa[i], wv);
            }
        }
#pragma unroll
        for (int i = 0; i < S::RTW; ++i)
#pragma unroll
            for (int j = 0; j < S::HCW; ++j) {
                const int col = (wc * S::HCW + j) * 16 + r;
                const float bias = a.b ? a.b[col] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) s_y[((wr * S::RTW + i) * 16 + q * 4 + e) * S::LD + col] = acc[i][j][e] + bias;
            }
        __syncthreads();
        for (int i = tid; i < kTileRows * (M / 4); i += kThreads) {
            const int row = i / (M / 4), c4 = (i % (M / 4)) * 4;
            const int64_t node = base + row;
            if (node < a.N) st4(a.Y + node * a.ldy + c4, ld4(s_y + row * S::LD + c4));
        }
        // s_x is rewritten by the next iteration only after every wave has passed the barrier above;
        // s_y is rewritten only after the next iteration's first barrier
    }
}

// dW[M][K] += dY^T [X1|X2],  db[M] += colsum(dY).  Each wave reduces its own 16 rows of the tile into
// (M/16)*(K/16) register tiles that persist across the workgroup's tiles; one atomic flush at the end.
template <int M, int K>
__global__ __launch_bounds__(kThreads) void k_linear_wgrad(LinArgs a) {
    constexpr int MC = M / 16, KC = K / 16, LDX = K + 4, LDY = M + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_x = smem;
    float* s_dy = smem + kTileRows * LDX;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    f32x4 acc[MC][KC];
#pragma unroll
    for (int i = 0; i < MC; ++i)
#pragma unroll
        for (int j = 0; j < KC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int64_t ntiles = (a.N + kTileRows - 1) / kTileRows;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * kTileRows;
        __syncthreads();
        for (int i = tid; i < kTileRows * (K / 4); i += kThreads) {
            const int row = i / (K / 4), c4 = (i % (K / 4)) * 4;
            const int64_t node = base + row;
            float4 v = zero4();
            if (node < a.N) v = c4 < a.K1 ? ld4(a.X1 + node * a.ld1 + c4) : ld4(a.X2 + node * a.ld2 + (c4 - a.K1));
            st4(s_x + row * LDX + c4, v);
        }
        for (int i = tid; i < kTileRows * (M / 4); i += kThreads) {
            const int row = i / (M / 4), c4 = (i % (M / 4)) * 4;
            const int64_t node = base + row;
            st4(s_dy + row * LDY + c4, node < a.N ? ld4(a.dY + node * a.lddy + c4) : zero4());
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int row = 16 * w + kk * 4 + q;
            float av[MC], bv[KC];
#pragma unroll
            for (int i = 0; i < MC; ++i) av[i] = s_dy[row * LDY + i * 16 + r];
#pragma unroll
            for (int j = 0; j < KC; ++j) bv[j] = s_x[row * LDX + j * 16 + r];
#pragma unroll
            for (int i = 0; i < MC; ++i)
#pragma unroll
                for (int j = 0; j < KC; ++j) acc[i][j] = mfma16(av[i], bv[j], acc[i][j]);
        }
        if (a.db && tid < M) {
            float s = 0.f;
            for (int row = 0; row < kTileRows; ++row) s += s_dy[row * LDY + tid];
            bsum += s;
        }
    }
#pragma unroll
    for (int i = 0; i < MC; ++i)
#pragma unroll
        for (int j = 0; j < KC; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(a.dW + (int64_t)(i * 16 + q * 4 + e) * K + j * 16 + r, acc[i][j][e]);
    if (a.db && tid < M) atomicAdd(a.db + tid, bsum);
}

// agg[i] = sum_{j in nbr(i)} h[j],  deg[i] = |nbr(i)|
template <int H>
__global__ __launch_bounds__(kThreads) void k_gather_sum(int64_t N, const float* h, const int32_t* ptr, const int32_t* idx,
                                                        float* agg, float* deg) {
    constexpr int LPR = H / 4;
    const int64_t gid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kThreads / LPR;
    const int lr = threadIdx.x % LPR;
    for (int64_t node = gid / LPR; node < N; node += stride) {
        const int e0 = ptr[node], e1 = ptr[node + 1];
        float4 acc = zero4();
        for (int e = e0; e < e1; ++e) acc = add4(acc, ld4(h + (int64_t)idx[e] * H + 4 * lr));
        st4(agg + node * H + 4 * lr, acc);
        if (deg && lr == 0) deg[node] = (float)(e1 - e0);
    }
}

// First half round of an encoder: every node starts from the same state (ones), so a node's output row depends
// only on its (degree, feature class) pair.  Forward = the stage kernel on one representative row per pair + this
// expansion; backward = per-pair sums of the incoming gradient (with the neighbour pull fused) + the stage backward
// on the representative rows (parameter gradients are linear in the incoming gradient).
template <int H>
__global__ __launch_bounds__(kThreads) void k_class_expand(int64_t N, const float* table, const int32_t* class_id, float* out) {
    constexpr int LPR = H / 4;
    const int64_t stride = (int64_t)gridDim.x * kThreads / LPR;
    const int lr = threadIdx.x % LPR;
    for (int64_t node = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / LPR; node < N; node += stride)
        st4(out + node * H + 4 * lr, ld4(table + (int64_t)class_id[node] * H + 4 * lr));
}

// CREG > 0: at most CREG classes, every lane keeps one partial sum per class in registers (the LDS atomic unit
// would otherwise serialise: most rows of a wave share a class) and adds them to LDS once at the end
template <int H, int CREG>
__global__ __launch_bounds__(kThreads) void k_class_pull_sum(int64_t N, const float* gy_direct, const float* gy_agg, const int32_t* ptr,
                                                            const int32_t* idx, const int32_t* class_id, int C, float* slab) {
    constexpr int LPR = H / 4;
    extern __shared__ __attribute__((aligned(16))) float s_acc[];        // [C][H], then (CREG > 0) the per-wave stage [waves][C][H]
    float4 racc[CREG > 0 ? CREG : 1];
#pragma unroll
    for (int c = 0; c < (CREG > 0 ? CREG : 1); ++c) racc[c] = zero4();
    for (int i = threadIdx.x; i < C * H; i += kThreads) s_acc[i] = 0.f;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kThreads / LPR;
    const int lr = threadIdx.x % LPR;
    constexpr int U = 4, D = 2;          // rows per lane group in flight, neighbours per row gathered together
    for (int64_t node0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / LPR; node0 < N; node0 += U * stride) {
        int e0[U], e1[U], cls[U];
        f32x4 own[U], nb[U][D];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t node = node0 + u * stride;
            e0[u] = e1[u] = 0; cls[u] = 0;
            if (node < N) {
                own[u] = *reinterpret_cast<const f32x4*>(gy_direct + node * H + 4 * lr);
                cls[u] = class_id[node];
                if (gy_agg) { e0[u] = ptr[node]; e1[u] = ptr[node + 1]; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (e0[u] + k < e1[u]) nb[u][k] = *reinterpret_cast<const f32x4*>(gy_agg + (int64_t)idx[e0[u] + k] * H + 4 * lr);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t node = node0 + u * stride;
            if (node >= N) continue;
            float4 acc = make_float4(own[u][0], own[u][1], own[u][2], own[u][3]);
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (e0[u] + k < e1[u]) acc = add4(acc, make_float4(nb[u][k][0], nb[u][k][1], nb[u][k][2], nb[u][k][3]));
            for (int e = e0[u] + D; e < e1[u]; ++e) acc = add4(acc, ld4(gy_agg + (int64_t)idx[e] * H + 4 * lr));
            if (CREG > 0) {
#pragma unroll
                for (int c = 0; c < CREG; ++c) {
                    const float m = cls[u] == c ? 1.0f : 0.0f;
                    racc[c] = fma4(m, acc, racc[c]);
                }
            } else {
                float* d = s_acc + cls[u] * H + 4 * lr;
                atomicAdd(d + 0, acc.x); atomicAdd(d + 1, acc.y); atomicAdd(d + 2, acc.z); atomicAdd(d + 3, acc.w);
            }
        }
    }
    // per-workgroup result to this workgroup's slab row; the rows are added in a fixed order afterwards (mgv_slab.h)
    float* row = slab + (int64_t)blockIdx.x * C * H;
    if (CREG > 0) {
        // no float atomics: the lane groups of a wave meet by shuffles (fixed tree), the waves through an LDS stage in wave order
        constexpr int NWV = kThreads / 64;
        float* s_stage = s_acc + C * H;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < CREG; ++c)
            if (c < C) {
                float4 v = racc[c];
#pragma unroll
                for (int mk = LPR; mk < 64; mk <<= 1) {
                    v.x += __shfl_xor(v.x, mk, 64); v.y += __shfl_xor(v.y, mk, 64); v.z += __shfl_xor(v.z, mk, 64); v.w += __shfl_xor(v.w, mk, 64);
                }
                if (lane < LPR) st4(s_stage + (w * C + c) * H + 4 * lane, v);
            }
        __syncthreads();
        for (int i = threadIdx.x; i < C * H; i += kThreads) {
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < NWV; ++ww) v += s_stage[ww * C * H + i];
            row[i] = v;
        }
    } else {
        __syncthreads();
        for (int i = threadIdx.x; i < C * H; i += kThreads) row[i] = s_acc[i];      // (LDS float atomics above: order-dependent inside a workgroup)
    }
}

// Segmented row sums, one lane group per segment, members added in list order (deterministic):
//     out[s] = sum over m in [seg_ptr[s], seg_ptr[s+1]) of v(item(m)),   item(m) = items ? items[m] : m,
//     v(i) = direct[i] + (agg ? sum over e in [nbr_ptr[i], nbr_ptr[i+1]) of agg[nbr_idx[e]] : 0).
// The per-class sums of an incoming gradient (quotient stages of the structural encoder, ops.StructEncoderFn): level 1 sums runs
// of <= 64 class members (with the neighbour pull of the stage's backward fused), further levels sum the partial rows; every
// level's segments are cut class by class on the host side (GraphPlan.class_sum_levels), so no float atomics and no order left
// to chance.  U members' loads are in flight together; the additions keep list order.
#ifndef MGV_SEG_U
#define MGV_SEG_U 2          // members of a segment in flight together
#define MGV_SEG_D 2          // neighbour rows per member in flight together
#endif
template <int H>
__global__ __launch_bounds__(kThreads) void k_seg_sum(int64_t n_seg, const int32_t* seg_ptr, const int32_t* items, const float* direct,
                                                     const float* agg, const int32_t* ptr, const int32_t* idx, const int32_t* out_row, float* out) {
    constexpr int LPR = H / 4, U = MGV_SEG_U, D = MGV_SEG_D;
    const int lr = threadIdx.x % LPR;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / LPR);
    for (int64_t s = (int64_t)blockIdx.x * (kThreads / LPR) + threadIdx.x / LPR; s < n_seg; s += stride) {
        const int m0 = seg_ptr[s], m1 = seg_ptr[s + 1];
        float4 acc = zero4();
        for (int mb = m0; mb < m1; mb += U) {
            int64_t row[U];
            int e0[U], e1[U];
            f32x4 own[U], nb[U][D];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                row[u] = -1; e0[u] = e1[u] = 0;
                if (mb + u < m1) {
                    row[u] = items ? (int64_t)items[mb + u] : (int64_t)(mb + u);
                    own[u] = *reinterpret_cast<const f32x4*>(direct + row[u] * H + 4 * lr);
                    if (agg) { e0[u] = ptr[row[u]]; e1[u] = ptr[row[u] + 1]; }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < D; ++k)
                    if (e0[u] + k < e1[u]) nb[u][k] = *reinterpret_cast<const f32x4*>(agg + (int64_t)idx[e0[u] + k] * H + 4 * lr);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (row[u] < 0) continue;
                float4 v = make_float4(own[u][0], own[u][1], own[u][2], own[u][3]);
#pragma unroll
                for (int k = 0; k < D; ++k)
                    if (e0[u] + k < e1[u]) v = add4(v, make_float4(nb[u][k][0], nb[u][k][1], nb[u][k][2], nb[u][k][3]));
                for (int e = e0[u] + D; e < e1[u]; ++e) v = add4(v, ld4(agg + (int64_t)idx[e] * H + 4 * lr));
                acc = add4(acc, v);
            }
        }
        st4(out + (out_row ? (int64_t)out_row[s] : s) * H + 4 * lr, acc);
    }
}

template <int M>
int launch_linear_fwd(const LinArgs& a, hipStream_t st) {
    const int K = a.K1 + a.K2;
    const size_t shm = (size_t)kTileRows * ((K + 4) + (M + 4)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_fwd<M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    const int64_t ntiles = (a.N + kTileRows - 1) / kTileRows;
    hipLaunchKernelGGL(k_linear_fwd<M>, dim3(grid_for(ntiles, 8)), dim3(kThreads), shm, st, a);
    MGV_LAUNCH_RET();
}

template <int M, int K>
int launch_linear_wgrad(const LinArgs& a, hipStream_t st) {
    const size_t shm = (size_t)kTileRows * ((K + 4) + (M + 4)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_wgrad<M, K>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    const int64_t ntiles = (a.N + kTileRows - 1) / kTileRows;
    hipLaunchKernelGGL((k_linear_wgrad<M, K>), dim3(grid_for(ntiles, 2)), dim3(kThreads), shm, st, a);
    MGV_LAUNCH_RET();
}

}  // namespace mgv

extern "C" int mgv_linear_fwd(int64_t N, const float* X1, int K1, int ld1, const float* X2, int K2, int ld2,
                              const float* W, const float* b, int M, float* Y, int ldy, void* stream) {
    MGV_CHECK_ARG(N >= 0 && X1 && W && Y && K1 > 0 && K2 >= 0 && (K2 == 0 || X2));
    MGV_CHECK_ARG(K1 % 4 == 0 && K2 % 4 == 0 && (K1 + K2) % 16 == 0 && (K1 + K2) <= 256);
    MGV_CHECK_ARG(ld1 >= K1 && ld1 % 4 == 0 && (K2 == 0 || (ld2 >= K2 && ld2 % 4 == 0)) && ldy >= M && ldy % 4 == 0);
    if (N == 0) return MGV_OK;
    mgv::LinArgs a{};
    a.N = N; a.X1 = X1; a.K1 = K1; a.ld1 = ld1; a.X2 = X2; a.K2 = K2; a.ld2 = ld2; a.W = W; a.b = b; a.Y = Y; a.ldy = ldy;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (M) {
        case 16: return mgv::launch_linear_fwd<16>(a, st);
        case 32: return mgv::launch_linear_fwd<32>(a, st);
        case 64: return mgv::launch_linear_fwd<64>(a, st);
        case 128: return mgv::launch_linear_fwd<128>(a, st);
        default: return MGV_EUNSUPPORTED;
    }
}

extern "C" int mgv_linear_wgrad(int64_t N, const float* X1, int K1, int ld1, const float* X2, int K2, int ld2,
                                const float* dY, int lddy, int M, float* dW, float* db, void* stream) {
    MGV_CHECK_ARG(N >= 0 && X1 && dY && dW && K1 > 0 && K2 >= 0 && (K2 == 0 || X2));
    MGV_CHECK_ARG(K1 % 4 == 0 && K2 % 4 == 0 && ld1 % 4 == 0 && (K2 == 0 || ld2 % 4 == 0) && lddy % 4 == 0 && lddy >= M);
    if (N == 0) return MGV_OK;
    mgv::LinArgs a{};
    a.N = N; a.X1 = X1; a.K1 = K1; a.ld1 = ld1; a.X2 = X2; a.K2 = K2; a.ld2 = ld2; a.dY = dY; a.lddy = lddy; a.dW = dW; a.db = db;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int K = K1 + K2;
#define MGV_WG(MM, KK) if (M == MM && K == KK) return mgv::launch_linear_wgrad<MM, KK>(a, st);
    MGV_WG(16, 16) MGV_WG(16, 32) MGV_WG(32, 16) MGV_WG(32, 32) MGV_WG(32, 64) MGV_WG(64, 16) MGV_WG(64, 32)
    MGV_WG(64, 64) MGV_WG(64, 128) MGV_WG(128, 64)
#undef MGV_WG
    return MGV_EUNSUPPORTED;
}

extern "C" int mgv_gather_sum(int H, int64_t N, const float* h, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                              float* agg, float* deg, void* stream) {
    MGV_CHECK_ARG(N >= 0 && h && nbr_ptr && agg);
    if (N == 0) return MGV_OK;
    MGV_CHECK_ARG(nbr_idx != nullptr);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t rows_per_block = mgv::kThreads / (H / 4);
    const int grid = mgv::grid_for((N + rows_per_block - 1) / rows_per_block, 16);
    switch (H) {
        case 16: hipLaunchKernelGGL(mgv::k_gather_sum<16>, dim3(grid), dim3(mgv::kThreads), 0, st, N, h, nbr_ptr, nbr_idx, agg, deg); break;
        case 32: hipLaunchKernelGGL(mgv::k_gather_sum<32>, dim3(grid), dim3(mgv::kThreads), 0, st, N, h, nbr_ptr, nbr_idx, agg, deg); break;
        case 64: hipLaunchKernelGGL(mgv::k_gather_sum<64>, dim3(grid), dim3(mgv::kThreads), 0, st, N, h, nbr_ptr, nbr_idx, agg, deg); break;
        case 128: hipLaunchKernelGGL(mgv::k_gather_sum<128>, dim3(grid), dim3(mgv::kThreads), 0, st, N, h, nbr_ptr, nbr_idx, agg, deg); break;
        default: return MGV_EUNSUPPORTED;
    }
    MGV_LAUNCH_RET();
}

extern "C" int mgv_seg_sum(int H, int64_t n_seg, const int32_t* seg_ptr, const int32_t* items, const float* direct, const float* agg,
                           const int32_t* nbr_ptr, const int32_t* nbr_idx, const int32_t* out_row, float* out, void* stream) {
    MGV_CHECK_ARG(n_seg >= 0 && seg_ptr && direct && out && (!agg || (nbr_ptr && nbr_idx)));
    if (n_seg == 0) return MGV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t rows_per_block = mgv::kThreads / (H / 4);
    const int grid = mgv::grid_for((n_seg + rows_per_block - 1) / rows_per_block, 16);
    switch (H) {
        case 16: hipLaunchKernelGGL(mgv::k_seg_sum<16>, dim3(grid), dim3(mgv::kThreads), 0, st, n_seg, seg_ptr, items, direct, agg, nbr_ptr, nbr_idx, out_row, out); break;
        case 32: hipLaunchKernelGGL(mgv::k_seg_sum<32>, dim3(grid), dim3(mgv::kThreads), 0, st, n_seg, seg_ptr, items, direct, agg, nbr_ptr, nbr_idx, out_row, out); break;
        case 64: hipLaunchKernelGGL(mgv::k_seg_sum<64>, dim3(grid), dim3(mgv::kThreads), 0, st, n_seg, seg_ptr, items, direct, agg, nbr_ptr, nbr_idx, out_row, out); break;
        case 128: hipLaunchKernelGGL(mgv::k_seg_sum<128>, dim3(grid), dim3(mgv::kThreads), 0, st, n_seg, seg_ptr, items, direct, agg, nbr_ptr, nbr_idx, out_row, out); break;
        default: return MGV_EUNSUPPORTED;
    }
    MGV_LAUNCH_RET();
}

extern "C" int mgv_class_expand(int H, int64_t N, const float* table, const int32_t* class_id, float* out, void* stream) {
    MGV_CHECK_ARG(N >= 0 && table && class_id && out);
    if (N == 0) return MGV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows_per_block = mgv::kThreads / (H / 4);
    const int grid = mgv::grid_for((N + rows_per_block - 1) / rows_per_block, 16);
    switch (H) {
        case 16: hipLaunchKernelGGL(mgv::k_class_expand<16>, dim3(grid), dim3(mgv::kThreads), 0, st, N, table, class_id, out); break;
        case 32: hipLaunchKernelGGL(mgv::k_class_expand<32>, dim3(grid), dim3(mgv::kThreads), 0, st, N, table, class_id, out); break;
        case 64: hipLaunchKernelGGL(mgv::k_class_expand<64>, dim3(grid), dim3(mgv::kThreads), 0, st, N, table, class_id, out); break;
        case 128: hipLaunchKernelGGL(mgv::k_class_expand<128>, dim3(grid), dim3(mgv::kThreads), 0, st, N, table, class_id, out); break;
        default: return MGV_EUNSUPPORTED;
    }
    MGV_LAUNCH_RET();
}

static int class_pull_grid(int H, int64_t N) {
    const int rows_per_block = mgv::kThreads / (H / 4);
    return mgv::grid_for((N + rows_per_block - 1) / rows_per_block, 8);
}

// floats of workspace mgv_class_pull_sum needs: one row of C*H partial sums per workgroup
extern "C" int mgv_class_pull_sum_ws_floats(int H, int64_t N, int C) {
    if (N <= 0 || C <= 0 || H <= 0) return 0;
    return class_pull_grid(H, N) * C * H;
}

extern "C" int mgv_class_pull_sum(int H, int64_t N, const float* gy_direct, const float* gy_agg, const int32_t* nbr_ptr,
                                  const int32_t* nbr_idx, const int32_t* class_id, int C, float* out, float* workspace,
                                  int64_t workspace_floats, void* stream) {
    MGV_CHECK_ARG(N >= 0 && gy_direct && class_id && out && C >= 1 && (int64_t)C * H * 4 * 5 <= 160 * 1024 && (!gy_agg || (nbr_ptr && nbr_idx)));
    if (N == 0) return MGV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = class_pull_grid(H, N);
    MGV_CHECK_ARG(workspace && workspace_floats >= (int64_t)grid * C * H);
    const size_t shm = (size_t)C * H * sizeof(float) * (C <= 8 ? 1 + mgv::kThreads / 64 : 1);
#define MGV_CPS(HH) \
    case HH: { \
        if (C <= 8) { \
            static bool set = false; \
            if (!set) { hipFuncSetAttribute(reinterpret_cast<const void*>(mgv::k_class_pull_sum<HH, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; } \
            hipLaunchKernelGGL((mgv::k_class_pull_sum<HH, 8>), dim3(grid), dim3(mgv::kThreads), shm, st, N, gy_direct, gy_agg, nbr_ptr, nbr_idx, class_id, C, workspace); \
        } else { \
            static bool set0 = false; \
            if (!set0) { hipFuncSetAttribute(reinterpret_cast<const void*>(mgv::k_class_pull_sum<HH, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set0 = true; } \
            hipLaunchKernelGGL((mgv::k_class_pull_sum<HH, 0>), dim3(grid), dim3(mgv::kThreads), shm, st, N, gy_direct, gy_agg, nbr_ptr, nbr_idx, class_id, C, workspace); \
        } \
        break; }
    switch (H) {
        MGV_CPS(16) MGV_CPS(32) MGV_CPS(64) MGV_CPS(128)
        default: return MGV_EUNSUPPORTED;
    }
#undef MGV_CPS
    mgv::launch_slab_sum<float, float>(workspace, grid, (int64_t)C * H, C * H, out, st);
    MGV_LAUNCH_RET();
}
