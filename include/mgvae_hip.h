/* mgvae_hip.h — C ABI of libmgvae_hip.so: the MI355X (gfx950) kernels behind the DG_AE hot path.
 *
 * The reference (959AI994/Multi-Gate-VAE) has no FFI of its own: its hot path is Python calling
 * ATen / PyG operators.  Each entry point below names the reference operator(s) it replaces
 * (file:line under /root/reference/DG_VAE/deepgate/) — these are the calls a maintainer would bind
 * (ctypes stub: INTEGRATION.md).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless said otherwise; the library allocates nothing, keeps
 *     no state between calls and only enqueues work on `stream` (a hipStream_t passed as void*);
 *   - return value: 0 = enqueued, MGV_EINVAL (-1) = bad argument, MGV_EUNSUPPORTED (-2) = unsupported
 *     size (hidden width H must be 16, 32 or 64), > 0 = hipError_t of the failed launch;
 *   - matrices are row-major fp32, node indices int32, all "gradient accumulator" outputs (dW..,
 *     db..) are ADDED to with atomics: the caller zeroes them;
 *   - N = nodes in the batch, E = edges, H = dim_hidden, gate column blocks are ordered r,z,n like
 *     torch.nn.GRU.
 */
#ifndef MGVAE_HIP_H
#define MGVAE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#ifndef MGV_OK
#define MGV_OK 0
#define MGV_EINVAL (-1)
#define MGV_EUNSUPPORTED (-2)
#endif

int mgv_abi_version(void);

/* ---- structural encoder half round: AggConv -> GRU -> LayerNorm
 * replaces digae_layer.py:267-270 (forward edges) / :272-275 (reversed edges) with
 * arch/gcn_conv.py:30-45, torch.nn.GRU (seq_len 1) and torch.nn.LayerNorm.
 *   nbr_ptr[N+1], nbr_idx[E] : CSR of the nodes each node sums over (in-neighbours for the forward
 *                               half, out-neighbours for the reversed half)
 *   xcls[N], xtab[C][3H]      : feature-row class per node and W_ih[:,H:] x_c + b_ih per class
 *   Wc[3H][H], bc[3H]         : W_ih[:,:H] Wm and W_ih[:,:H] bm (message Linear folded into the GRU)
 *   ln_w/ln_b                 : NULL,NULL = no LayerNorm */
int mgv_struct_stage_fwd(int H, int64_t N, const float* h_in, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                         const uint8_t* xcls, const float* xtab, int C, const float* Wc, const float* bc,
                         const float* Whh, const float* bhh, const float* ln_w, const float* ln_b, float ln_eps,
                         float* h_out, void* stream);
/* backward of the above (what autograd does for the same lines).  The incoming gradient is
 * dY[i] = gy_direct[i] + sum_{j in nbr(i)} gy_agg[j]  (gy_agg may be NULL): consecutive half rounds
 * use opposite CSRs, so the scatter of the NEXT stage's aggregate gradient is this stage's gather.
 * Outputs: g_direct_out = dL/dh_in through the GRU's hidden path, g_agg_out = dL/d(sum of neighbour
 * rows) (both NULL when h_in is a constant); WcT/WhhT are the [H][3H] transposes. */
int mgv_struct_stage_bwd(int H, int64_t N, const float* h_in, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                         const uint8_t* xcls, const float* xtab, int C, const float* Wc, const float* WcT,
                         const float* bc, const float* Whh, const float* WhhT, const float* bhh,
                         const float* ln_w, const float* ln_b, float ln_eps, const float* gy_direct,
                         const float* gy_agg, float* g_direct_out, float* g_agg_out, float* dWc, float* dbc,
                         float* dWhh, float* dbhh, float* dxtab, float* dln_w, float* dln_b, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGVAE_HIP_H */
