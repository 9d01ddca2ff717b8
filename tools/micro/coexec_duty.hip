// Micro-benchmark, round 4: VALU / MFMA co-execution of the two waves of a SIMD at PARTIAL matrix duty.
// tools/micro/coexec.hip drove the matrix pipe at 100 % (back-to-back independent v_mfma_f32_16x16x32_bf16) and found
// "both = sum".  The struct-stage kernels run the pipe at ~27 %.  Here the MFMA wave issues ONE MFMA every GAP cycles
// (GAP = 16: back to back, 32: 50 % duty, 64: 25 % duty; the gap is s_nop filler in the SAME wave, so the SIMD's issue port
// is free for its partner in between), the partner wave issues dependent-free v_fma_f32 (or v_pk_fma_f32 / v_exp_f32) all the time.
// Timed with HIP events: MFMA waves alone, vector waves alone, both.  If vector instructions of the partner can issue under a
// running MFMA, "both" ~ max(alone); if the SIMD is held for the MFMA's issue cycles, both ~ vector alone + (MFMA count x 16 cycles).
// (s_nop 15 is 16 QUAD-cycles = 64 cycles: the cases are one MFMA per 16 / 80 / 208 / 464 cycles = 100 / 20 / 8 / 3.5 % duty.)
// The vector waves also time THEIR OWN loop with s_memtime: beside a partial-duty MFMA wave the wall time is the MFMA wave's, so
// whether the vector wave was held up shows only in its own cycles (alone vs beside the MFMA wave).
// Every (GAP, KIND, WHO) is its own kernel instantiation, so `rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
// SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU -- tools/bin/coexec_duty` reports the counters per case.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void valu_op(f32x2& v, const f32x2& m, const f32x2& ad) {
    if (KIND == 0) v = v * m + ad;                                              // v_pk_fma_f32
    else if (KIND == 1) { v[0] = __builtin_fmaf(v[0], m[0], ad[0]); }            // v_fma_f32
    else { v[0] = __builtin_amdgcn_exp2f(v[0]); }                                // v_exp_f32
}

template <int GAP, int KIND, int WHO>      // WHO bit 0: the MFMA waves (0-3) work, bit 1: the vector waves (4-7) work
__global__ __launch_bounds__(512) void k_duty(float* out, int iters, unsigned long long* vcycles) {
    const int w = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
    f32x4 c[4];
    for (int j = 0; j < 4; ++j) c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 v[8];
    for (int j = 0; j < 8; ++j) v[j] = f32x2{0.001f * threadIdx.x + j, 0.5f};
    const f32x2 m = f32x2{1.0001f, 0.9999f}, ad = f32x2{1e-6f, -1e-6f};
    if (w < 4 && (WHO & 1)) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u & 3], 0, 0, 0);
                if (GAP >= 32) asm volatile("s_nop 15");
                if (GAP >= 64) asm volatile("s_nop 15\n\ts_nop 15");
                if (GAP >= 128) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
            }
    }
    if (w >= 4 && (WHO & 2)) {
        unsigned long long t0, t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 24 * 4; ++u) valu_op<KIND>(v[u & 7], m, ad);            // 96 vector instructions per iteration
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(v[0][0]), "v"(v[7][0]) : "memory");
        if ((threadIdx.x & 63) == 0) atomicAdd(vcycles, t1 - t0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][3];
    for (int j = 0; j < 8; ++j) s += v[j][0] + v[j][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int GAP, int KIND, int WHO>
float run(float* out, int iters, double* vec_cycles_per_inst = nullptr) {
    static unsigned long long* vc = nullptr;
    if (!vc) hipMalloc(&vc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_duty<GAP, KIND, WHO><<<256, 512>>>(out, iters, vc);
    hipMemset(vc, 0, 8);
    hipEventRecord(e0);
    k_duty<GAP, KIND, WHO><<<256, 512>>>(out, iters, vc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long tot = 0;
    hipMemcpy(&tot, vc, 8, hipMemcpyDeviceToHost);
    if (vec_cycles_per_inst) *vec_cycles_per_inst = (double)tot / (256.0 * 4.0) / ((double)iters * 96.0);      // per vector wave and instruction
    return ms;
}

template <int GAP, int KIND>
void report(float* out, const char* name, int iters) {
    double va = 0, vb = 0;
    const float a = run<GAP, KIND, 1>(out, iters), b = run<GAP, KIND, 2>(out, iters, &va), c = run<GAP, KIND, 3>(out, iters, &vb);
    const int period = GAP == 16 ? 16 : GAP == 32 ? 80 : GAP == 64 ? 208 : 464;
    printf("one MFMA per %3d cycles (%5.1f %% duty), partner %-12s: wall: MFMA waves alone %.2f ms, vector waves alone %.2f ms, both %.2f ms | "
           "the vector wave's own s_memtime ticks per instruction: alone %.3f, beside the MFMA wave %.3f (x%.3f; exclusive issue predicts x%.3f)\n",
           period, 1600.0f / period, name, a, b, c, va, vb, vb / va, 1.0 / (1.0 - 16.0 / period));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float* out; hipMalloc(&out, 256 * 512 * 4);
    report<16, 1>(out, "v_fma_f32", iters); report<32, 1>(out, "v_fma_f32", iters); report<64, 1>(out, "v_fma_f32", iters); report<128, 1>(out, "v_fma_f32", iters);
    report<16, 0>(out, "v_pk_fma_f32", iters); report<32, 0>(out, "v_pk_fma_f32", iters); report<64, 0>(out, "v_pk_fma_f32", iters);
    report<32, 2>(out, "v_exp_f32", iters); report<64, 2>(out, "v_exp_f32", iters);
    hipFree(out);
    return 0;
}
