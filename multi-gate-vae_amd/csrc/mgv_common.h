// Shared device helpers for the MI355X (gfx950 / CDNA4) kernels of the DG_AE hot path.
// Wave = 64 lanes everywhere; MFMA = v_mfma_f32_16x16x4_f32 (exact f32, k-ordered fmaf chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MGV_OK 0
#define MGV_EINVAL (-1)      // bad argument (null pointer, unsupported H, negative size)
#define MGV_EUNSUPPORTED (-2)

#define MGV_CHECK_ARG(cond) do { if (!(cond)) return MGV_EINVAL; } while (0)
#define MGV_LAUNCH_RET() do { hipError_t e__ = hipGetLastError(); return e__ == hipSuccess ? MGV_OK : (int)e__; } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace mgv {

constexpr int kWave = 64;
constexpr int kThreads = 256;     // 4 waves, one per SIMD of a CU
constexpr int kTileRows = 64;     // nodes per workgroup tile

// v_exp_f32 + v_rcp_f32 (1 ulp each): absolute error ~1e-7, no IEEE division sequence, no branches
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// 4-vector forms: the multiplies / adds around the transcendental instructions compile to packed fp32 operations
__device__ __forceinline__ f32x4 ldv4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 sigmoid4(const f32x4& x) {
    const f32x4 t = x * -1.4426950408889634f;
    const f32x4 d = f32x4{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])} + 1.0f;
    return f32x4{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
}
__device__ __forceinline__ f32x4 tanh4(const f32x4& x) {
    const f32x4 t = x * 2.8853900817779268f;
    const f32x4 d = f32x4{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])} + 1.0f;
    const f32x4 r = f32x4{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
    return 1.0f - 2.0f * r;
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One 16-deep k-block of D(16x16) += X(16xK) * W(16xK)^T where both operands are row-major with k
// contiguous.  Lane (r = lane&15, q = lane>>4) loads k = kb+4q..kb+4q+3 of its row of X and of its
// row of W as one float4; the four MFMAs then each see the SAME k on the A and the B side (a
// consistent permutation of k inside the block, which a dot product does not care about).
__device__ __forceinline__ void mma_kblock(f32x4& acc, const float4& a, const float4& b) {
    acc = mfma16(a.x, b.x, acc);
    acc = mfma16(a.y, b.y, acc);
    acc = mfma16(a.z, b.z, acc);
    acc = mfma16(a.w, b.w, acc);
}

// Cross-lane reductions.  hipcc lowers __shfl_xor to ds_bpermute_b32 (an LDS-pipe round trip per step);
// inside a 16-lane row the same butterfly is four v_add_f32_dpp: quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror, row_mirror (after the quad steps every lane of a quad holds the quad's sum, so the
// mirrors pair complementary quads / halves).  Only the 32- and 64-wide tails go through bpermute.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// sum over the `width` (power of two <= 64) consecutive lanes a lane's group consists of
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
    if (WIDTH >= 2) v += dpp_mov<0xB1>(v);
    if (WIDTH >= 4) v += dpp_mov<0x4E>(v);
    if (WIDTH >= 8) v += dpp_mov<0x141>(v);
    if (WIDTH >= 16) v += dpp_mov<0x140>(v);
    if (WIDTH >= 32) v += __shfl_xor(v, 16, 64);
    if (WIDTH >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
    if (WIDTH >= 2) v = fmaxf(v, dpp_mov<0xB1>(v));
    if (WIDTH >= 4) v = fmaxf(v, dpp_mov<0x4E>(v));
    if (WIDTH >= 8) v = fmaxf(v, dpp_mov<0x141>(v));
    if (WIDTH >= 16) v = fmaxf(v, dpp_mov<0x140>(v));
    if (WIDTH >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (WIDTH >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

// block-wide sum for 256-thread blocks; result valid in thread 0 (and all of wave 0)
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats of LDS */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float s, const float4& a, const float4& b) { return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w)); }
__device__ __forceinline__ float4 scale4(float s, const float4& a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// How the 4 waves of a workgroup split a (64 rows) x (H hidden columns) output whose three GRU gate
// column blocks must land in the same lane: waves go across hidden-column tiles first (so each wave
// streams only its slice of the weights), then across row tiles.
template <int H>
struct WaveSplit {
    static_assert(H % 16 == 0 && H >= 16 && H <= 128, "H must be a multiple of 16 in [16,128]");
    static constexpr int HC = H / 16;                  // hidden column tiles
    static constexpr int WPC = HC < 4 ? HC : 4;        // waves across column tiles
    static constexpr int WPR = 4 / WPC;                // waves across row tiles
    static constexpr int RTW = 4 / WPR;                // row tiles (of 16) per wave
    static constexpr int HCW = HC / WPC;               // column tiles per wave
    static constexpr int LD = H + 4;                   // padded LDS leading dimension (floats)
    static constexpr int LPR = H / 4;                  // lanes per row when a row is read as float4s
    static constexpr int GROUPS = kThreads / LPR;      // rows processed concurrently by the block
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup release/acquire fence: with
// LDS-DMA loads in flight (the L2 prefetches) the compiler drains vmcnt(0) in front of it, which would expose the
// very latency the prefetch is there to hide; no kernel here exchanges data through global memory inside a tile.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

inline int grid_for(int64_t tiles, int per_cu) {
    int64_t cap = 256LL * per_cu;
    return (int)(tiles < cap ? (tiles < 1 ? 1 : tiles) : cap);
}

}  // namespace mgv
