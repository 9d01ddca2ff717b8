// Micro-benchmark: v_mfma_f32_16x16x32_bf16 issued as ONE dependent accumulation chain vs 2 / 4 independent chains,
// one wave per SIMD and two waves per SIMD.  Prints cycles per MFMA per wave (s_memtime around the loop, lane 0 of wave 0).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
    f32x4 c[CH];
    for (int j = 0; j < CH; ++j) c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24 / CH; ++u)
#pragma unroll
            for (int j = 0; j < CH; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < CH; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH>
void run(int threads, const char* tag) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CH><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e0);
    k<CH><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = 256.0 * (threads / 64) * iters * 24.0 * 16384.0;
    printf("%-14s chains=%d: %.1f s_memtime ticks per MFMA per wave; wall %.3f ms -> %.0f TFLOP/s; implied clock %.2f GHz\n", tag, CH, (double)h / (iters * 24.0), ms,
           flops / (ms * 1e-3) / 1e12, (double)h / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>(256, "1 wave/SIMD"); run<2>(256, "1 wave/SIMD"); run<3>(256, "1 wave/SIMD"); run<4>(256, "1 wave/SIMD"); run<8>(256, "1 wave/SIMD");
    run<1>(512, "2 waves/SIMD"); run<2>(512, "2 waves/SIMD"); run<4>(512, "2 waves/SIMD");
    return 0;
}
