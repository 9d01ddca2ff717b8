// Deterministic sums across workgroups.  A kernel whose workgroups each hold a partial of the same small vector (a weight
// gradient, a column sum, a loss sum) stores it with plain stores to its own row of a caller-provided slab [workgroups][stride];
// k_slab_sum then adds the rows in a FIXED order into the destination (+=).  No float atomics: two identical calls give
// bit-identical results, and the order does not depend on which workgroup finished first.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mgv {

// small-sum launchers (column statistics, loss sums, head gradients): at most kSumRows workgroups, at most kSumStride doubles each
constexpr int kSumRows = 2048;
constexpr int kSumStride = 136;

// 64 outputs per block, 16 row phases per output (rows ty, ty+16, ...; eight loads in flight per thread), LDS combine in phase
// order: the result depends on the rows' contents only, never on which workgroup wrote its row first
template <typename T, typename O>
static __global__ __launch_bounds__(1024) void k_slab_sum(const T* slab, int nwg, int64_t stride, int n, O* out) {
    constexpr int P = 16;
    __shared__ T red[P][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    T s = 0;
    if (i < n) {
        const T* src = slab + i;
        int g = ty;
        for (; g + 7 * P < nwg; g += 8 * P) {
            T v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[(int64_t)(g + k * P) * stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; g < nwg; g += P) s += src[(int64_t)g * stride];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n) {
        T t = red[0][tx];
#pragma unroll
        for (int k = 1; k < P; ++k) t += red[k][tx];
        out[i] += (O)t;
    }
}

template <typename T, typename O>
static inline void launch_slab_sum(const T* slab, int nwg, int64_t stride, int n, O* out, hipStream_t st) {
    if (n <= 0 || nwg <= 0) return;
    hipLaunchKernelGGL((k_slab_sum<T, O>), dim3((n + 63) / 64), dim3(1024), 0, st, slab, nwg, stride, n, out);
}

// out[nodes[k]][0..H) (+)= sum of the node's segment partials partial[seg][H] in segment order (heavy-list kernels: a node whose
// list is too long for one lane group is cut into segments, one workgroup each)
static __global__ __launch_bounds__(256) void k_heavy_add(int K, int H, const int32_t* nodes, const int32_t* node_seg_ptr, const float* partial,
                                                          float* out, int ld, int accumulate) {
    const int lpr = H / 4, lr = threadIdx.x % lpr;
    for (int k = blockIdx.x * (256 / lpr) + threadIdx.x / lpr; k < K; k += gridDim.x * (256 / lpr)) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sg = node_seg_ptr[k]; sg < node_seg_ptr[k + 1]; ++sg) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)sg * H + 4 * lr);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float4* dst = reinterpret_cast<float4*>(out + (int64_t)nodes[k] * ld + 4 * lr);
        if (accumulate) { const float4 o = *dst; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        *dst = s;
    }
}

// out[k][0..W) = sum of node k's segment partials (k in [k0, k1): rows indexed by k, not by node id)
static __global__ __launch_bounds__(256) void k_heavy_add_range(int k0, int k1, int W, const int32_t* node_seg_ptr, const float* partial, float* out) {
    const int lpr = W / 4, lr = threadIdx.x % lpr;
    for (int k = k0 + blockIdx.x * (256 / lpr) + threadIdx.x / lpr; k < k1; k += gridDim.x * (256 / lpr)) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sg = node_seg_ptr[k]; sg < node_seg_ptr[k + 1]; ++sg) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)sg * W + 4 * lr);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(out + (int64_t)k * W + 4 * lr) = s;
    }
}

}  // namespace mgv
