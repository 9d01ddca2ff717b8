// Stand-alone TFMlpAggr (arch/tfmlp.py:31-46) outside the levelised sweep: attention pooling of source rows over an edge list given
// as a CSR by destination.
//     zbar[i] = sum_j alpha_ij x_j,   alpha_i. = softmax_j(u . x_j)   (torch_geometric softmax: exp(s - max) / (sum exp + 1e-16))
// The message of the module is then W_v zbar + b_v [deg > 0] (the linear kernels): sum_j alpha_j (W_v x_j + b_v) with sum alpha = 1.
// Same restatement as the sweep kernels: the q term of the score is constant over a destination's softmax segment and cancels.
// One lane group (W/4 lanes) per destination node, online softmax (one pass over the sources), fp32 throughout.
// Backward: d(score_j) in the centred form alpha_j dz . (x_j - zbar); the rows' gradients are scattered with float atomics (this
// entry is not on the train step: inside a Model the aggregation runs in the sweep kernels, whose backward pulls instead).
#include "mgv_common.h"
#include "../../include/mgvae_hip.h"

namespace mgv {

template <int W>
__global__ __launch_bounds__(kThreads) void k_attn_pool_fwd(int64_t N, const int32_t* ptr, const int32_t* idx, const float* x, const float* u,
                                                           float* zbar, float* mstat, float* inv) {
    constexpr int LPR = W / 4;
    const int lr = threadIdx.x % LPR;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / LPR);
    const float4 uu = ld4(u + 4 * lr);
    for (int64_t node = (int64_t)blockIdx.x * (kThreads / LPR) + threadIdx.x / LPR; node < N; node += stride) {
        const int e0 = ptr[node], e1 = ptr[node + 1];
        float m = -INFINITY, S = 0.f;
        float4 z = zero4();
        for (int e = e0; e < e1; ++e) {
            const float4 xj = ld4(x + (int64_t)idx[e] * W + 4 * lr);
            const float sc = group_sum<LPR>(dot4(uu, xj));
            const float mn = fmaxf(m, sc);
            const float corr = __expf(m - mn), w = __expf(sc - mn);
            S = S * corr + w;
            z = fma4(w, xj, scale4(corr, z));
            m = mn;
        }
        const float iv = 1.0f / (S + 1e-16f);
        st4(zbar + node * W + 4 * lr, scale4(iv, z));
        if (lr == 0) { mstat[node] = e1 > e0 ? m : 0.f; inv[node] = iv; }
    }
}

template <int W>
__global__ __launch_bounds__(kThreads) void k_attn_pool_bwd(int64_t N, const int32_t* ptr, const int32_t* idx, const float* x, const float* u,
                                                           const float* zbar, const float* mstat, const float* inv, const float* dz,
                                                           float* dx, float* du) {
    constexpr int LPR = W / 4;
    __shared__ __attribute__((aligned(16))) float s_du[kThreads / LPR][W];
    const int lr = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const int64_t stride = (int64_t)gridDim.x * (kThreads / LPR);
    const float4 uu = ld4(u + 4 * lr);
    float4 gu = zero4();
    for (int64_t node = (int64_t)blockIdx.x * (kThreads / LPR) + grp; node < N; node += stride) {
        const int e0 = ptr[node], e1 = ptr[node + 1];
        if (e1 <= e0) continue;
        const float4 zb = ld4(zbar + node * W + 4 * lr), g = ld4(dz + node * W + 4 * lr);
        const float m = mstat[node], iv = inv[node];
        for (int e = e0; e < e1; ++e) {
            const int64_t j = idx[e];
            const float4 xj = ld4(x + j * W + 4 * lr);
            const float al = __expf(group_sum<LPR>(dot4(uu, xj)) - m) * iv;
            const float4 c = make_float4(xj.x - zb.x, xj.y - zb.y, xj.z - zb.z, xj.w - zb.w);
            const float ds = al * group_sum<LPR>(dot4(g, c));
            float* d = dx + j * W + 4 * lr;
            atomicAdd(d + 0, al * g.x + ds * uu.x); atomicAdd(d + 1, al * g.y + ds * uu.y);
            atomicAdd(d + 2, al * g.z + ds * uu.z); atomicAdd(d + 3, al * g.w + ds * uu.w);
            gu = fma4(ds, xj, gu);
        }
    }
    st4(&s_du[grp][4 * lr], gu);
    __syncthreads();
    for (int c = threadIdx.x; c < W; c += kThreads) {
        float s = 0.f;
        for (int g2 = 0; g2 < kThreads / LPR; ++g2) s += s_du[g2][c];
        atomicAdd(du + c, s);
    }
}

}  // namespace mgv

#define MGV_DISPATCH_W(W, CALL)                          \
    switch (W) {                                         \
        case 32: { constexpr int WW = 32; CALL; } break;   \
        case 64: { constexpr int WW = 64; CALL; } break;   \
        case 128: { constexpr int WW = 128; CALL; } break; \
        default: return MGV_EUNSUPPORTED;                \
    }

extern "C" int mgv_attn_pool_fwd(int W, int64_t N, const int32_t* in_ptr, const int32_t* in_src, const float* x, const float* u,
                                 float* zbar, float* mstat, float* inv, void* stream) {
    MGV_CHECK_ARG(N >= 0 && in_ptr && x && u && zbar && mstat && inv);
    if (N == 0) return MGV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows_per_block = mgv::kThreads / (W / 4);
    const int grid = mgv::grid_for((N + rows_per_block - 1) / rows_per_block, 8);
    MGV_DISPATCH_W(W, hipLaunchKernelGGL((mgv::k_attn_pool_fwd<WW>), dim3(grid), dim3(mgv::kThreads), 0, st, N, in_ptr, in_src, x, u, zbar, mstat, inv));
    MGV_LAUNCH_RET();
}

extern "C" int mgv_attn_pool_bwd(int W, int64_t N, const int32_t* in_ptr, const int32_t* in_src, const float* x, const float* u,
                                 const float* zbar, const float* mstat, const float* inv, const float* dzbar, float* dx, float* du,
                                 void* stream) {
    MGV_CHECK_ARG(N >= 0 && in_ptr && x && u && zbar && mstat && inv && dzbar && dx && du);
    if (N == 0) return MGV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows_per_block = mgv::kThreads / (W / 4);
    const int grid = mgv::grid_for((N + rows_per_block - 1) / rows_per_block, 8);
    MGV_DISPATCH_W(W, hipLaunchKernelGGL((mgv::k_attn_pool_bwd<WW>), dim3(grid), dim3(mgv::kThreads), 0, st, N, in_ptr, in_src, x, u, zbar, mstat, inv, dzbar, dx, du));
    MGV_LAUNCH_RET();
}
