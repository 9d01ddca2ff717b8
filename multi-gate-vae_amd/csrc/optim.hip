// Adam over one flat fp32 parameter buffer (torch.optim.Adam semantics, trainer.py:73), with the
// gradient optionally pre-scaled (1/world_size after the RCCL all-reduce sum).
#include "mgv_common.h"
#include "../../include/mgvae_hip.h"

namespace mgv {

__global__ __launch_bounds__(kThreads) void k_adam(int64_t n, float* p, const float* g, float* m, float* v, float lr, float b1,
                                                   float b2, float eps, float wd, float gscale, float bc1, float bc2_sqrt) {
    const float step = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step * (mi / denom);
    }
}

}  // namespace mgv

extern "C" int mgv_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                             float beta1, float beta2, float eps, float weight_decay, float grad_scale, int64_t step, void* stream) {
    MGV_CHECK_ARG(n >= 0 && step >= 1);
    if (n == 0) return MGV_OK;
    MGV_CHECK_ARG(param && grad && exp_avg && exp_avg_sq);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(mgv::k_adam, dim3(mgv::grid_for((n + mgv::kThreads - 1) / mgv::kThreads, 8)), dim3(mgv::kThreads), 0,
                       static_cast<hipStream_t>(stream), n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay,
                       grad_scale, (float)bc1, (float)sqrt(bc2));
    MGV_LAUNCH_RET();
}
