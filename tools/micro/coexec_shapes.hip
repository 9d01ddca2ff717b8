// Micro-benchmark: does any bf16 MFMA shape leave issue slots for the SIMD partner's vector instructions?
// Same set-up as coexec.hip (waves 0-3 MFMA, waves 4-7 v_fma_f32), per MFMA shape: MFMA alone, vector alone, both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mfma_on, int valu_on) {
    const int w = threadIdx.x >> 6;
    bf16x8 a8, b8;
    s16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(0.001f * (threadIdx.x + i)); b8[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
    for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)(threadIdx.x * 3 + i); }
    f32x4 c4[4];
    f32x16 c16[2];
    for (int j = 0; j < 4; ++j) c4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) c16[j][e] = 0.f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.001f * threadIdx.x + j;
    if (w < 4 && mfma_on) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                if (SHAPE == 0) c4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c4[u & 3], 0, 0, 0);
                else if (SHAPE == 1) c4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c4[u & 3], 0, 0, 0);
                else if (SHAPE == 2) c16[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, c16[u & 1], 0, 0, 0);
                else if (SHAPE == 3) c16[u & 1] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, c16[u & 1], 0, 0, 0);
            }
    }
    if (w >= 4 && valu_on) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 96; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], 1.0001f, 1e-6f);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += c4[j][0] + c4[j][3];
    for (int j = 0; j < 2; ++j) s += c16[j][0] + c16[j][15];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE>
float run(int mfma_on, int valu_on) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE><<<256, 512>>>(out, 20000, mfma_on, valu_on);
    hipEventRecord(e0);
    k<SHAPE><<<256, 512>>>(out, 20000, mfma_on, valu_on);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}
template <int SHAPE>
void report(const char* name, double flop) {
    const float a = run<SHAPE>(1, 0), b = run<SHAPE>(0, 1), c = run<SHAPE>(1, 1);
    printf("%-28s MFMA alone %.2f ms (%.0f TFLOP/s at one wave per SIMD), v_fma alone %.2f ms, both %.2f ms (sum %.2f, max %.2f)\n", name, a,
           256.0 * 4 * 20000 * 24 * flop / (a * 1e-3) / 1e12, b, c, a + b, a > b ? a : b);
}
int main() {
    for (int i = 0; i < 20; ++i) run<0>(1, 1);      // clocks up before the first row
    report<0>("v_mfma_f32_16x16x32_bf16", 16384.0); report<1>("v_mfma_f32_16x16x16_bf16", 8192.0);
    report<2>("v_mfma_f32_32x32x16_bf16", 32768.0); report<3>("v_mfma_f32_32x32x8_bf16", 16384.0);
    return 0;
}
