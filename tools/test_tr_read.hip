// Hardware check of ds_read_b64_tr_b16 semantics (exact small integers in bf16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int LD = 72;
__global__ void k(const float* in, float* out) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[16 * LD];
    for (int i = threadIdx.x; i < 16 * LD; i += 64) tile[i] = (__bf16)in[i];
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    // every 16-lane group g reads the block rows 4g..4g+3, cols 0..15
    __bf16* addr = tile + (4 * g + q) * LD + 4 * p;
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)addr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (float)v[e];
}
int main() {
    std::vector<float> h(16 * LD, 0.f), o(256);
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) h[r * LD + c] = r * 16 + c;
    float *di, *dout;
    hipMalloc(&di, h.size() * 4); hipMalloc(&dout, 256 * 4);
    hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o.data(), dout, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
        int g = lane >> 4, i = lane & 15;
        float want = (4 * g + e) * 16 + i;          // lane i of group g: column i, rows 4g..4g+3 in elements 0..3
        if (o[lane * 4 + e] != want) { if (bad < 8) printf("lane %d e %d got %g want %g\n", lane, e, o[lane * 4 + e], want); ++bad; }
    }
    printf(bad ? "TR_READ MISMATCH %d\n" : "TR_READ OK (lane i <- column i, element e <- row e of the 4x16 block)\n", bad);
    return bad != 0;
}
