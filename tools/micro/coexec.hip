// Micro-benchmark: can the two waves of a SIMD co-execute, one issuing v_mfma_f32_16x16x32_bf16 and the other vector FMAs?
// A workgroup of 8 waves: waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 (their SIMD partners) a packed-FMA loop.
// Timed: MFMA waves alone, VALU waves alone, both together (wall time of 256 workgroups, one per CU).  If the pipes overlap, "both"
// costs max(alone); if they take turns, the sum.  Third mode: ONE wave interleaving both instruction kinds (independent streams).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void valu_op(f32x2& v, const f32x2& m, const f32x2& ad) {
    if (KIND == 0) v = v * m + ad;                                              // v_pk_fma_f32
    else if (KIND == 1) { v[0] = __builtin_fmaf(v[0], m[0], ad[0]); }            // v_fma_f32
    else if (KIND == 2) { unsigned u = __builtin_bit_cast(unsigned, v[0]); u = u * 3u + 7u; v[0] = __builtin_bit_cast(float, u); }   // integer mad
    else if (KIND == 3) { v[0] = __builtin_amdgcn_exp2f(v[0]); }                   // v_exp_f32 (quarter rate)
    else if (KIND == 4) { unsigned u = __builtin_bit_cast(unsigned, v[0]); u ^= 0x9e3779b9u; u = (u << 3) | (u >> 29); v[0] = __builtin_bit_cast(float, u); }   // logic / shift
}

template <int MODE, int KIND>      // MODE 0: partner waves (mask selects who works), 1: one wave interleaves MFMA and the vector op
__global__ __launch_bounds__(512) void k(float* out, int iters, int mfma_on, int valu_on, int valu_per_mfma) {
    const int w = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
    f32x4 c[4];
    for (int j = 0; j < 4; ++j) c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 v[8];
    for (int j = 0; j < 8; ++j) v[j] = f32x2{0.001f * threadIdx.x + j, 0.5f};
    const f32x2 m = f32x2{1.0001f, 0.9999f}, ad = f32x2{1e-6f, -1e-6f};
    if (MODE == 0) {
        if (w < 4 && mfma_on) {
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int u = 0; u < 24; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u & 3], 0, 0, 0);
        }
        if (w >= 4 && valu_on) {
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int u = 0; u < 24 * 4; ++u) valu_op<KIND>(v[u & 7], m, ad);        // 96 vector instructions per iteration (4 per MFMA slot)
        }
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u & 3], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 3; ++q) valu_op<KIND>(v[(3 * u + q) & 7], m, ad);
            }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][3];
    for (int j = 0; j < 8; ++j) s += v[j][0] + v[j][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int KIND>
float run(int threads, int mfma_on, int valu_on) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, KIND><<<256, threads>>>(out, iters, mfma_on, valu_on, 4);
    hipEventRecord(e0);
    k<MODE, KIND><<<256, threads>>>(out, iters, mfma_on, valu_on, 4);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}
template <int KIND>
void report(const char* name) {
    const float a = run<0, KIND>(512, 1, 0), b = run<0, KIND>(512, 0, 1), c = run<0, KIND>(512, 1, 1), d = run<1, KIND>(256, 1, 1);
    printf("%-22s partner waves: MFMA alone %.2f ms, vector alone %.2f ms (96 per 24 MFMA), both %.2f ms (sum %.2f, max %.2f) | one wave, 3 per MFMA interleaved: %.2f ms (sum %.2f)\n",
           name, a, b, c, a + b, a > b ? a : b, d, a + 0.75f * b);
}
int main() {
    report<0>("v_pk_fma_f32"); report<1>("v_fma_f32"); report<2>("integer mul-add"); report<3>("v_exp_f32"); report<4>("xor / shift / or");
    return 0;
}
