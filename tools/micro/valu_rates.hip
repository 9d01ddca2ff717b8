// Micro-benchmark: issue cost of the vector instructions the struct kernels use most, one wave per SIMD, 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.001f * threadIdx.x + j + 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 96; ++u) {
            float& x = v[u & 7];
            if (KIND == 0) x = __builtin_fmaf(x, 1.0001f, 1e-6f);
            else if (KIND == 1) { f32x2 p = f32x2{x, x} * f32x2{1.0001f, 0.9999f} + f32x2{1e-6f, 1e-6f}; x = p[0] + 0.f * p[1]; }
            else if (KIND == 2) { bf16x2 h = bf16x2{(__bf16)x, (__bf16)(x * 0.5f)}; x = x + 1e-6f * (float)h[0] + 0.f * (float)h[1]; }      // v_cvt_pk_bf16_f32 (+ unpack)
            else if (KIND == 3) x = __builtin_amdgcn_exp2f(x * 1e-3f);
            else if (KIND == 4) x = __builtin_amdgcn_rcpf(x + 1.5f);
            else if (KIND == 5) x = x + __shfl_xor(x, 16, 64) * 1e-6f;                                      // cross-lane (ds_bpermute / dpp / permlane)
            else if (KIND == 6) { unsigned uu = __builtin_bit_cast(unsigned, x); uu = (uu & 0xffff0000u) | (uu >> 16); x = __builtin_bit_cast(float, uu) * 1e-30f + 1.0f; }
        }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name, float base) {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, 256>>>(out, iters);
    hipEventRecord(e0);
    k<KIND><<<256, 256>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.3f ms for 96 x 20000 per wave = %.2f ns each\n", name, ms, ms * 1e6 / (96.0 * iters));
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32", 0); run<1>("v_pk_fma_f32 (+ 1 add)", 0); run<2>("v_cvt_pk_bf16_f32 (+ unpack, fma)", 0); run<3>("v_exp_f32 (+ mul)", 0);
    run<4>("v_rcp_f32 (+ add)", 0); run<5>("__shfl_xor 16 (+ fma)", 0); run<6>("and / shift / or pack (+ fma)", 0);
    return 0;
}
