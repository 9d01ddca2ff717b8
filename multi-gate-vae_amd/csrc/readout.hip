// Readout MLP pieces around the MFMA Linear kernels (mlp.py:27-47, dg_ae_model_aig.py:102-106):
// BatchNorm1d batch statistics, BN + ReLU + Dropout, the 32->1 head with clamp and the L1 loss,
// and their backward passes.  All streaming over [N, C] (C = dim_mlp = 32), C/4 lanes per row.
#include "mgv_common.h"
#include "mgv_slab.h"
#include "../../include/mgvae_hip.h"

namespace mgv {

constexpr int kMaxC = 64;

__device__ __forceinline__ uint32_t hash_u32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}
// inverted-dropout factor of element (row, col): 0 or 1/(1-p); the same counter-based stream is
// regenerated in the backward pass
__device__ __forceinline__ float drop_scale(uint64_t seed, int64_t elem, float p, float keep_scale) {
    if (p <= 0.f) return 1.0f;
    const uint32_t h = hash_u32(seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(elem + 1));
    const float u = (h >> 8) * (1.0f / 16777216.0f);
    return u < p ? 0.f : keep_scale;
}

// sums[c] += sum_i Y[i][c],  sums[C + c] += sum_i Y[i][c]^2   (double)
__global__ __launch_bounds__(kThreads) void k_colstats(int64_t N, int C, const float* Y, int ld, double* slab) {     // slab [gridDim][2C]
    double* sums = slab + (int64_t)blockIdx.x * 2 * C;
    __shared__ double red[2 * kThreads];
    const int cq = C / 4, rows = kThreads / cq;
    const int c4 = (threadIdx.x % cq) * 4, r0 = threadIdx.x / cq;
    double s[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (r0 < rows)
        for (int64_t i = (int64_t)blockIdx.x * rows + r0; i < N; i += (int64_t)gridDim.x * rows) {
            const float4 v = ld4(Y + i * ld + c4);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            s2[0] += (double)v.x * v.x; s2[1] += (double)v.y * v.y; s2[2] += (double)v.z * v.z; s2[3] += (double)v.w * v.w;
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        __syncthreads();
        red[threadIdx.x] = s[k]; red[kThreads + threadIdx.x] = s2[k];
        __syncthreads();
        if (threadIdx.x < cq) {
            double a = 0, b = 0;
            for (int r = 0; r < rows; ++r) { a += red[r * cq + threadIdx.x]; b += red[kThreads + r * cq + threadIdx.x]; }
            sums[threadIdx.x * 4 + k] = a;          // this workgroup's row; rows are added in a fixed order afterwards
            sums[C + threadIdx.x * 4 + k] = b;
        }
    }
}

// A = dropout(relu(gamma * (Y - mean) * invstd + beta))
__global__ __launch_bounds__(kThreads) void k_bn_act_fwd(int64_t N, int C, const float* Y, const float* mean, const float* invstd,
                                                         const float* gamma, const float* beta, float p, uint64_t seed, float* A) {
    const int cq = C / 4;
    const int64_t total = N * cq;
    const float ks = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int c4 = (int)(i % cq) * 4;
        const float4 y = ld4(Y + i * 4), m = ld4(mean + c4), is = ld4(invstd + c4), g = ld4(gamma + c4), b = ld4(beta + c4);
        float4 o;
        o.x = fmaxf((y.x - m.x) * is.x * g.x + b.x, 0.f) * drop_scale(seed, i * 4 + 0, p, ks);
        o.y = fmaxf((y.y - m.y) * is.y * g.y + b.y, 0.f) * drop_scale(seed, i * 4 + 1, p, ks);
        o.z = fmaxf((y.z - m.z) * is.z * g.z + b.z, 0.f) * drop_scale(seed, i * 4 + 2, p, ks);
        o.w = fmaxf((y.w - m.w) * is.w * g.w + b.w, 0.f) * drop_scale(seed, i * 4 + 3, p, ks);
        st4(A + i * 4, o);
    }
}

// dZ = dA * dropout_scale * [bn_out > 0];  sums[c] += sum dZ, sums[C+c] += sum dZ * xhat
__global__ __launch_bounds__(kThreads) void k_bn_act_bwd(int64_t N, int C, const float* Y, const float* mean, const float* invstd,
                                                         const float* gamma, const float* beta, float p, uint64_t seed,
                                                         const float* dA, float* dZ, double* slab) {     // slab [gridDim][2C]
    double* sums = slab + (int64_t)blockIdx.x * 2 * C;
    __shared__ double red[2 * kThreads];
    const int cq = C / 4, rows = kThreads / cq;
    const int c4 = (threadIdx.x % cq) * 4, r0 = threadIdx.x / cq;
    const float ks = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    double s[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (r0 < rows) {
        const float4 m = ld4(mean + c4), is = ld4(invstd + c4), g = ld4(gamma + c4), b = ld4(beta + c4);
        const float mm[4] = {m.x, m.y, m.z, m.w}, ii[4] = {is.x, is.y, is.z, is.w}, gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
        const int64_t stride = (int64_t)gridDim.x * rows;
        constexpr int U = 4;             // rows in flight per thread; their sums meet in fp32, then join the double totals
        for (int64_t i0 = (int64_t)blockIdx.x * rows + r0; i0 < N; i0 += U * stride) {
            float4 y[U], da[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (i0 + u * stride < N) { y[u] = ld4(Y + (i0 + u * stride) * C + c4); da[u] = ld4(dA + (i0 + u * stride) * C + c4); }
            float fs[4] = {0.f, 0.f, 0.f, 0.f}, fs2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (i0 + u * stride < N) {
                    const int64_t i = i0 + u * stride;
                    const float yy[4] = {y[u].x, y[u].y, y[u].z, y[u].w}, dd[4] = {da[u].x, da[u].y, da[u].z, da[u].w};
                    float o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float xhat = (yy[k] - mm[k]) * ii[k];
                        const float bn = xhat * gg[k] + bb[k];
                        const float dz = bn > 0.f ? dd[k] * drop_scale(seed, i * C + c4 + k, p, ks) : 0.f;
                        o[k] = dz; fs[k] += dz; fs2[k] += dz * xhat;
                    }
                    st4(dZ + i * C + c4, make_float4(o[0], o[1], o[2], o[3]));
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k] += fs[k]; s2[k] += fs2[k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        __syncthreads();
        red[threadIdx.x] = s[k]; red[kThreads + threadIdx.x] = s2[k];
        __syncthreads();
        if (threadIdx.x < cq) {
            double a = 0, b = 0;
            for (int r = 0; r < rows; ++r) { a += red[r * cq + threadIdx.x]; b += red[kThreads + r * cq + threadIdx.x]; }
            sums[threadIdx.x * 4 + k] = a;          // this workgroup's row; rows are added in a fixed order afterwards
            sums[C + threadIdx.x * 4 + k] = b;
        }
    }
}

// dY = gamma * invstd * (dZ - [batch stats] (S1/N + xhat * S2/N))
__global__ __launch_bounds__(kThreads) void k_bn_bwd_apply(int64_t N, int C, const float* Y, const float* mean, const float* invstd,
                                                           const float* gamma, const float* dZ, const double* sums, int batch_stats, float* dY) {
    const int cq = C / 4;
    const int64_t total = N * cq;
    const double invn = 1.0 / (double)N;
    // kThreads % cq == 0: a thread keeps its column quad over the grid-stride loop, so the per-column constants are
    // formed once (they were re-read and re-multiplied in double per element)
    const int c4 = (int)(threadIdx.x % cq) * 4;
    float mm[4], sc[4], a1[4], a2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c4 + k;
        const float is = invstd[c];
        mm[k] = mean[c];
        sc[k] = gamma[c] * is;
        a1[k] = batch_stats ? (float)(sums[c] * invn) : 0.f;
        a2[k] = batch_stats ? (float)(sums[C + c] * invn) * is : 0.f;       // multiplies (y - mean)
    }
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    constexpr int U = 4;
    for (int64_t i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x; i0 < total; i0 += U * stride) {
        float4 y[U], dz[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u * stride < total) { y[u] = ld4(Y + (i0 + u * stride) * 4); dz[u] = ld4(dZ + (i0 + u * stride) * 4); }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u * stride < total) {
                const float yy[4] = {y[u].x, y[u].y, y[u].z, y[u].w}, zz[4] = {dz[u].x, dz[u].y, dz[u].z, dz[u].w};
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = sc[k] * (zz[k] - a1[k] - (yy[k] - mm[k]) * a2[k]);
                st4(dY + (i0 + u * stride) * 4, make_float4(o[0], o[1], o[2], o[3]));
            }
    }
}

// prob = clamp(A w + b, 0, 1)
// BWD: dy = dprob * [0 <= y <= 1]; dA = dy w; dw += sum dy A; db += sum dy
template <bool BWD>
__global__ __launch_bounds__(kThreads) void k_head(int64_t N, int C, const float* A, const float* w, const float* b, int clamp01,
                                                   float* prob, const float* dprob, float* dA, double* slab) {     // BWD: slab [gridDim][C + 1]
    const int cq = C / 4, rows = kThreads / cq;
    const int lr = threadIdx.x % cq, r0 = threadIdx.x / cq;
    const float4 wv = ld4(w + 4 * lr);
    const float bias = b[0];
    float4 dwv = zero4();
    float dbv = 0.f;
    const int64_t nblk = (N + rows - 1) / rows;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t i = blk * rows + r0;
        const bool ok = i < N;
        float4 a = zero4();
        if (ok) a = ld4(A + i * C + 4 * lr);
        float y = dot4(a, wv);
        // the C/4 lanes of one row are consecutive lanes of a wave (C/4 is a power of two <= 16)
        if (cq >= 2) y += dpp_mov<0xB1>(y);
        if (cq >= 4) y += dpp_mov<0x4E>(y);
        if (cq >= 8) y += dpp_mov<0x141>(y);
        if (cq >= 16) y += dpp_mov<0x140>(y);
        y += bias;
        if (!ok) continue;
        if (!BWD) {
            if (lr == 0) prob[i] = clamp01 ? fminf(fmaxf(y, 0.f), 1.f) : y;
        } else {
            const float dy = (!clamp01 || (y >= 0.f && y <= 1.f)) ? dprob[i] : 0.f;
            st4(dA + i * C + 4 * lr, scale4(dy, wv));
            dwv = fma4(dy, a, dwv);
            if (lr == 0) dbv += dy;
        }
    }
    if (BWD) {
        // no float atomics: rows of a wave that share a column quad meet by shuffles (fixed tree), the waves through LDS in wave
        // order; the workgroup's C + 1 partials go to its slab row (double) and the rows are added in a fixed order afterwards
        __shared__ float s_stage[kThreads / 64][kMaxC + 1];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int mk = cq; mk < 64; mk <<= 1) {
            dwv.x += __shfl_xor(dwv.x, mk, 64); dwv.y += __shfl_xor(dwv.y, mk, 64); dwv.z += __shfl_xor(dwv.z, mk, 64); dwv.w += __shfl_xor(dwv.w, mk, 64);
        }
        dbv = wave_sum(dbv);
        if (lane < cq) { s_stage[w][4 * lane + 0] = dwv.x; s_stage[w][4 * lane + 1] = dwv.y; s_stage[w][4 * lane + 2] = dwv.z; s_stage[w][4 * lane + 3] = dwv.w; }
        if (lane == 0) s_stage[w][kMaxC] = dbv;
        __syncthreads();
        double* row = slab + (int64_t)blockIdx.x * (C + 1);
        if (threadIdx.x <= C) {
            const int c = threadIdx.x < C ? threadIdx.x : kMaxC;
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < kThreads / 64; ++ww) v += s_stage[ww][c];
            row[threadIdx.x] = (double)v;
        }
    }
}

// nn.L1Loss (mean): sum += sum |x - t|;   dx = g/n * sign(x - t)
__global__ __launch_bounds__(kThreads) void k_l1_fwd(int64_t n, const float* x, const float* t, double* slab) {       // slab [gridDim]
    __shared__ double red[kThreads];
    double acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) acc += fabsf(x[i] - t[i]);
    __syncthreads();
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) slab[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(kThreads) void k_l1_bwd(int64_t n, const float* x, const float* t, const float* gscale, float* dx) {
    const float g = (*gscale) / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float d = x[i] - t[i];
        dx[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    }
}

inline int ew_grid(int64_t items) { return grid_for((items + kThreads - 1) / kThreads, 8); }
inline bool c_ok(int C) { return C == 4 || C == 8 || C == 16 || C == 32 || C == 64; }

}  // namespace mgv

// doubles of workspace the small-sum launchers need (mgv_colstats, mgv_bn_act_bwd, mgv_readout_head_bwd, mgv_l1_loss_fwd,
// mgv_recon_loss_fwd, mgv_func_loss_fwd): one row of partials per workgroup, added in a fixed order by a second launch
extern "C" int mgv_sum_workspace_doubles(void) { return mgv::kSumRows * mgv::kSumStride; }

extern "C" int mgv_colstats(int64_t N, int C, const float* Y, int ld, double* sums, double* workspace, int64_t workspace_doubles,
                            void* stream) {
    MGV_CHECK_ARG(N >= 0 && mgv::c_ok(C) && Y && sums && ld >= C && ld % 4 == 0);
    if (N == 0) return MGV_OK;
    const int rows = mgv::kThreads / (C / 4);
    const int grid = mgv::grid_for((N + rows - 1) / rows, 8);
    MGV_CHECK_ARG(workspace && workspace_doubles >= (int64_t)grid * 2 * C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mgv::k_colstats, dim3(grid), dim3(mgv::kThreads), 0, st, N, C, Y, ld, workspace);
    mgv::launch_slab_sum<double, double>(workspace, grid, 2 * C, 2 * C, sums, st);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_bn_act_fwd(int64_t N, int C, const float* Y, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, float p_drop, uint64_t seed, float* A, void* stream) {
    MGV_CHECK_ARG(N >= 0 && mgv::c_ok(C) && Y && mean && invstd && gamma && beta && A && p_drop >= 0.f && p_drop < 1.f);
    if (N == 0) return MGV_OK;
    hipLaunchKernelGGL(mgv::k_bn_act_fwd, dim3(mgv::ew_grid(N * (C / 4))), dim3(mgv::kThreads), 0, static_cast<hipStream_t>(stream),
                       N, C, Y, mean, invstd, gamma, beta, p_drop, seed, A);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_bn_act_bwd(int64_t N, int C, const float* Y, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, float p_drop, uint64_t seed, const float* dA, float* dZ, double* sums,
                              double* workspace, int64_t workspace_doubles, void* stream) {
    MGV_CHECK_ARG(N >= 0 && mgv::c_ok(C) && Y && mean && invstd && gamma && beta && dA && dZ && sums);
    if (N == 0) return MGV_OK;
    const int rows = mgv::kThreads / (C / 4);
    const int grid = mgv::grid_for((N + rows - 1) / rows, 8);
    MGV_CHECK_ARG(workspace && workspace_doubles >= (int64_t)grid * 2 * C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mgv::k_bn_act_bwd, dim3(grid), dim3(mgv::kThreads), 0, st, N, C, Y, mean, invstd, gamma, beta, p_drop, seed, dA, dZ, workspace);
    mgv::launch_slab_sum<double, double>(workspace, grid, 2 * C, 2 * C, sums, st);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_bn_bwd_apply(int64_t N, int C, const float* Y, const float* mean, const float* invstd, const float* gamma,
                                const float* dZ, const double* sums, int batch_stats, float* dY, void* stream) {
    MGV_CHECK_ARG(N >= 0 && mgv::c_ok(C) && Y && mean && invstd && gamma && dZ && sums && dY);
    if (N == 0) return MGV_OK;
    hipLaunchKernelGGL(mgv::k_bn_bwd_apply, dim3(mgv::ew_grid(N * (C / 4))), dim3(mgv::kThreads), 0, static_cast<hipStream_t>(stream),
                       N, C, Y, mean, invstd, gamma, dZ, sums, batch_stats, dY);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_readout_head_fwd(int64_t N, int C, const float* A, const float* w, const float* b, int clamp01, float* prob, void* stream) {
    MGV_CHECK_ARG(N >= 0 && mgv::c_ok(C) && A && w && b && prob);
    if (N == 0) return MGV_OK;
    const int rows = mgv::kThreads / (C / 4);
    hipLaunchKernelGGL((mgv::k_head<false>), dim3(mgv::grid_for((N + rows - 1) / rows, 8)), dim3(mgv::kThreads), 0,
                       static_cast<hipStream_t>(stream), N, C, A, w, b, clamp01, prob, nullptr, nullptr, nullptr);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_readout_head_bwd(int64_t N, int C, const float* A, const float* w, const float* b, int clamp01, const float* dprob,
                                    float* dA, float* dw, float* db, double* workspace, int64_t workspace_doubles, void* stream) {
    MGV_CHECK_ARG(N >= 0 && mgv::c_ok(C) && A && w && b && dprob && dA && dw && db);
    if (N == 0) return MGV_OK;
    const int rows = mgv::kThreads / (C / 4);
    const int grid = mgv::grid_for((N + rows - 1) / rows, 8);
    MGV_CHECK_ARG(workspace && workspace_doubles >= (int64_t)grid * (C + 1));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL((mgv::k_head<true>), dim3(grid), dim3(mgv::kThreads), 0, st, N, C, A, w, b, clamp01, nullptr, dprob, dA, workspace);
    mgv::launch_slab_sum<double, float>(workspace, grid, C + 1, C, dw, st);
    mgv::launch_slab_sum<double, float>(workspace + C, grid, C + 1, 1, db, st);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_l1_loss_fwd(int64_t n, const float* x, const float* target, double* sum, double* workspace,
                               int64_t workspace_doubles, void* stream) {
    MGV_CHECK_ARG(n >= 0 && sum);
    if (n == 0) return MGV_OK;
    MGV_CHECK_ARG(x && target);
    const int grid = mgv::ew_grid(n);
    MGV_CHECK_ARG(workspace && workspace_doubles >= grid);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mgv::k_l1_fwd, dim3(grid), dim3(mgv::kThreads), 0, st, n, x, target, workspace);
    mgv::launch_slab_sum<double, double>(workspace, grid, 1, 1, sum, st);
    MGV_LAUNCH_RET();
}

extern "C" int mgv_l1_loss_bwd(int64_t n, const float* x, const float* target, const float* gscale, float* dx, void* stream) {
    MGV_CHECK_ARG(n >= 0);
    if (n == 0) return MGV_OK;
    MGV_CHECK_ARG(x && target && gscale && dx);
    hipLaunchKernelGGL(mgv::k_l1_bwd, dim3(mgv::ew_grid(n)), dim3(mgv::kThreads), 0, static_cast<hipStream_t>(stream), n, x, target, gscale, dx);
    MGV_LAUNCH_RET();
}
