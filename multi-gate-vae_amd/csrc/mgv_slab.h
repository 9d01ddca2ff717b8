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

// 64 outputs per block, 4 row phases per output (rows ty, ty+4, ...), LDS combine in phase order
template <typename T, typename O>
static __global__ __launch_bounds__(256) void k_slab_sum(const T* slab, int nwg, int64_t stride, int n, O* out) {
    __shared__ T red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    T s = 0;
    if (i < n)
        for (int g = ty; g < nwg; g += 4) s += slab[(int64_t)g * stride + i];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n) out[i] += (O)(((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx]);
}

template <typename T, typename O>
static inline void launch_slab_sum(const T* slab, int nwg, int64_t stride, int n, O* out, hipStream_t st) {
    if (n <= 0 || nwg <= 0) return;
    hipLaunchKernelGGL((k_slab_sum<T, O>), dim3((n + 63) / 64), dim3(256), 0, st, slab, nwg, stride, n, out);
}

}  // namespace mgv
