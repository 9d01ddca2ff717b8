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
 *     db..) are ADDED to: the caller zeroes them.  On the default path (H = 64, bf16x3) the sums go through per-workgroup
 *     slabs and a fixed-order reduction kernel (bit-identical from run to run); the exact-fp32 family, H = 32 and the
 *     first bf16x3 backward use float atomics;
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
/* general node features (digae_layer.py:257-277 accepts any x [N, F]; the reference Models only ever feed one-hot rows): the same half
 * round with the GRU's feature term per NODE, xrow[N][3H] = W_ih[:, H:] x_i + b_ih (formed with mgv_linear_fwd), instead of the class
 * table.  Exact-fp32 kernels, H in {16, 32, 64}.  Backward: d_xrow[N][3H] is ADDED to (the caller sums it over the stages that
 * share the weights and carries it to W_ih[:, H:], b_ih and x through mgv_linear_*). */
int mgv_struct_stage_rows_fwd(int H, int64_t N, const float* h_in, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                              const float* xrow, const float* Wc, const float* bc, const float* Whh, const float* bhh,
                              const float* ln_w, const float* ln_b, float ln_eps, float* h_out, void* stream);
int mgv_struct_stage_rows_bwd(int H, int64_t N, const float* h_in, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                              const float* xrow, const float* Wc, const float* WcT, const float* bc, const float* Whh,
                              const float* WhhT, const float* bhh, const float* ln_w, const float* ln_b, float ln_eps,
                              const float* gy_direct, const float* gy_agg, float* g_direct_out, float* g_agg_out,
                              float* dWc, float* dbc, float* dWhh, float* dbhh, float* d_xrow, float* dln_w, float* dln_b,
                              void* stream);

/* ---- the same half round on bf16x3 split-precision MFMA (hi/lo bf16 planes, three products, fp32
 * accumulate; H in {32, 64}).  wpack_bf16 = eight bf16 blocks of 3H*H elements each:
 * Wc_hi, Wc_lo, Whh_hi, Whh_lo ([3H][H]) then WcT_hi, WcT_lo, WhhT_hi, WhhT_lo ([H][3H]);
 * hi = bf16(W), lo = bf16(W - hi).  All other arguments as for the fp32 entry points.
 * heavy_n / heavy_nodes / heavy_ws (0 / NULL / NULL when there are none): the nodes with more than 64 neighbours in this CSR
 * (a clock- or reset-like net), ascending, and 2 * heavy_n * H floats of scratch: their neighbour sums are formed by a pre-pass
 * with one workgroup per node instead of by one lane group inside the tile kernel (100,000 consumers: 38 ms per launch there).
 * table_own_idx (NULL = off): TABLE MODE for the half round that follows the (degree, class)-table one — h_in is the C-row table,
 * a node's own row is h_in[table_own_idx[node]], every nbr_idx entry carries its neighbour's table row in the top byte
 * (entry = node | row << 24, N < 2^24): the N x H expansion of the table is never gathered.  nbr_tagged = 0 with a table_own_idx:
 * the entries are plain rows of h_in and only the own rows go through the index (the quotient stages, GraphPlan.quotient: h_in is the
 * previous stage's colour table, a representative's own row is its previous colour's, its list names previous colours).
 * ln_stats_out (forward) / ln_stats (bwd2) [N][2], NULL = off: {mean, rstd} of every row's pre-LayerNorm state, kept by the forward
 * so that the backward's recompute needs two cross-lane row sums instead of four and no four-way combination of the column waves'
 * partial statistics (-3 % per backward launch); ignored when ln_w is NULL. */
int mgv_struct_stage_fwd_x3(int H, int64_t N, const float* h_in, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                            const uint8_t* xcls, const float* xtab, int C, const void* wpack_bf16, const float* bc,
                            const float* bhh, const float* ln_w, const float* ln_b, float ln_eps, float* h_out,
                            int heavy_n, const int32_t* heavy_nodes, float* heavy_ws, const int32_t* table_own_idx,
                            int nbr_tagged, float* ln_stats_out, void* stream);
int mgv_struct_stage_bwd_x3(int H, int64_t N, const float* h_in, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                            const uint8_t* xcls, const float* xtab, int C, const void* wpack_bf16, const float* bc,
                            const float* bhh, const float* ln_w, const float* ln_b, float ln_eps,
                            const float* gy_direct, const float* gy_agg, float* g_direct_out, float* g_agg_out,
                            float* dWc, float* dbc, float* dWhh, float* dbhh, float* dxtab, float* dln_w,
                            float* dln_b, int heavy_n, const int32_t* heavy_nodes, float* heavy_ws, const int32_t* table_own_idx,
                            int nbr_tagged, void* stream);

/* Second decomposition of the bf16x3 backward (H = 64 only): all weight fragments register-resident, transposed
 * products, one dgrad+wgrad phase per tile, and NO float atomics: parameter gradients leave through per-workgroup
 * slabs in `workspace` (mgv_struct_stage_bwd2_ws_floats(H, N) floats, device memory, contents irrelevant before and
 * after) and a fixed-order reduction, so two identical calls give bit-identical gradients.  Same reference lines and
 * argument meaning as mgv_struct_stage_bwd_x3 (digae_layer.py:266-275 under autograd). */
int mgv_struct_stage_bwd2_ws_floats(int H, int64_t N);   /* a size, not a status */
int mgv_struct_stage_bwd2_x3(int H, int64_t N, const float* h_in, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                             const uint8_t* xcls, const float* xtab, int C, const void* wpack_bf16, const float* bc,
                             const float* bhh, const float* ln_w, const float* ln_b, float ln_eps,
                             const float* gy_direct, const float* gy_agg, float* g_direct_out, float* g_agg_out,
                             float* dWc, float* dbc, float* dWhh, float* dbhh, float* dxtab, float* dln_w,
                             float* dln_b, float* workspace, int64_t workspace_floats, int heavy_n,
                             const int32_t* heavy_nodes, float* heavy_ws, const int32_t* table_own_idx,
                             int nbr_tagged, const float* ln_stats, void* stream);

/* ---- Linear over node rows (hs_linear dg_ae_model_aig.py:64, hs_decompose :109, fc_{s,t}_{mu,logstd}
 * digvae_model.py:135-136, readout Linear layers mlp.py:29,38; also the dgrad with W^T):
 *   Y[N][M] = [X1 | X2] W^T + b     (X2/K2 = NULL/0 unless a torch.cat of two inputs is fused, :64)
 * K1+K2 multiple of 16 (<= 256), M in {16,32,64,128}; ld* = row strides in floats. */
int mgv_linear_fwd(int64_t N, const float* X1, int K1, int ld1, const float* X2, int K2, int ld2,
                   const float* W, const float* b, int M, float* Y, int ldy, void* stream);
/* dW[M][K1+K2] += dY^T [X1|X2],  db[M] += column sums of dY (db may be NULL) */
int mgv_linear_wgrad(int64_t N, const float* X1, int K1, int ld1, const float* X2, int K2, int ld2,
                     const float* dY, int lddy, int M, float* dW, float* db, void* stream);
/* the same layers on bf16x3 split-precision MFMA (see mgv_struct_stage_fwd_x3): HBM-bound instead of fp32-MFMA-bound.
 * (M, K = K1 + K2) must be one of the shapes mgv_linear_x3_supported() accepts (the model's layer shapes at H=64/32);
 * forward weights arrive as wpack_bf16[2][M*K] = {W_hi, W_lo} in MFMA fragment order (see mgv_func_sweep_fwd_x3). */
int mgv_linear_x3_supported(int M, int K);
/* fragment-order bf16 hi/lo planes (R*K elements each) of the fp32 matrix A = W [R][K] (transpose = 0) or of the
 * transposed view A[i][k] = W[k][i] (transpose = 1; W is then [K][R]); ldw = leading dimension of W */
int mgv_wpack_bf16x3(const float* W, int R, int K, int ldw, int transpose, void* hi, void* lo, void* stream);
int mgv_linear_fwd_x3(int64_t N, const float* X1, int K1, int ld1, const float* X2, int K2, int ld2,
                      const void* wpack_bf16, const float* b, int M, float* Y, int ldy, void* stream);
/* grouped Linear over the level sweep's tiles (num_rounds > 1, dg_ae_model_aig.py:70,88-94: every updated gate's GRU adds W_hh h_prev + b_hh
 * with its OWN aggregator's weights — here one launch over all tiles instead of an index_select / Linear / index_copy per gate type).
 * Row r of tile t is NODE order[tile_start[t] + r] (r < tile_count[t]); X / Y / R / dY rows are indexed by node.  fwd: tile t multiplies by
 * the pack of slot tile_slot[t] (wpack_bf16[T][2][M*K], fragment order as mgv_wpack_bf16x3 writes it) and adds b[tile_slot[t]][M];
 * tile_list (nullable) names the tiles to visit (ntiles entries; NULL: tiles 0 .. ntiles-1).  wgrad: dW[M][K] += dY^T X, db[M] += colsum(dY)
 * over the rows of the listed tiles (ONE slot's list: GraphPlan.slot_tiles).  Shapes at H = 64: (M, K) = (192, 64) and (64, 192) forward,
 * (192, 64) weight gradient. */
int mgv_grouped_linear_supported(int M, int K);
int mgv_grouped_linear_fwd_x3(int64_t ntiles, const int32_t* tile_list, const int32_t* order, const int32_t* tile_start,
                              const int32_t* tile_count, const int32_t* tile_slot, const float* X, int K, int ldx,
                              const void* wpack_bf16, const float* b, int M, const float* R, int ldr, float* Y, int ldy, void* stream);
int mgv_grouped_linear_wgrad_x3_ws_floats(int M, int K, int64_t ntiles);                            /* a size */
int mgv_grouped_linear_wgrad_x3(int64_t ntiles, const int32_t* tile_list, const int32_t* order, const int32_t* tile_start,
                                const int32_t* tile_count, const float* X, int K, int ldx, const float* dY, int lddy, int M,
                                float* dW, float* db, float* workspace, int64_t workspace_floats, void* stream);
/* the same with residual rows: Y = [X1 | X2] W^T + b + R (R [N][ldr >= M]).  Used as the input gradient of a Linear whose input has a
 * second consumer (hs: hs_decompose and the level sweep, dg_ae_model_aig.py:64-70,109): the other consumer's gradient rides in as R
 * instead of meeting this one in a separate N x M add */
int mgv_linear_fwd_x3_res(int64_t N, const float* X1, int K1, int ld1, const float* X2, int K2, int ld2,
                          const void* wpack_bf16, const float* b, int M, const float* R, int ldr, float* Y, int ldy,
                          void* stream);
/* dW[M][K1+K2] += dY^T [X1 | X2], db[M] += column sums of dY (db nullable).  Deterministic: every workgroup leaves its partial in
 * its own row of `workspace` (at least mgv_linear_wgrad_x3_ws_floats(M, K1+K2, N) floats) and a second launch adds the rows in a
 * fixed order — no float atomics, bit-identical from call to call */
int mgv_linear_wgrad_x3_ws_floats(int M, int K, int64_t N);
int mgv_linear_wgrad_x3(int64_t N, const float* X1, int K1, int ld1, const float* X2, int K2, int ld2,
                        const float* dY, int lddy, int M, float* dW, float* db, float* workspace, int64_t workspace_floats,
                        void* stream);
/* agg[i] = sum_{j in nbr(i)} h[j], deg[i] = |nbr(i)| (deg may be NULL): the scatter-add half of
 * MessagePassing.propagate as used by AggConv called on its own (gcn_conv.py:34) */
int mgv_gather_sum(int H, int64_t N, const float* h, const int32_t* nbr_ptr, const int32_t* nbr_idx,
                   float* agg, float* deg, void* stream);

/* First half round of an encoder (digae_layer.py:260 starts every node from ones): an output row depends only on
 * the node's (degree, feature class) pair, class_id[N] numbers the pairs 0..C-1.
 * expand: out[i] = table[class_id[i]]; pull_sum: out[c] += sum over nodes of class c of
 * (gy_direct[i] + sum_{j in nbr(i)} gy_agg[j]) (gy_agg may be NULL); C * H * 20 <= 160 KiB.  pull_sum is deterministic for
 * C <= 8: per-workgroup rows in `workspace` (>= mgv_class_pull_sum_ws_floats(H, N, C) floats), added in a fixed order. */
int mgv_class_expand(int H, int64_t N, const float* table, const int32_t* class_id, float* out, void* stream);
/* Segmented row sums in list order (one lane group per segment, no atomics): out[s][H] = sum over m in [seg_ptr[s], seg_ptr[s+1]) of
 * v(item(m)), item(m) = items ? items[m] : m, v(i) = direct[i] + (agg ? sum of agg[nbr_idx[e]] over i's nbr list : 0).  The per-class
 * sums of an incoming gradient for the quotient stages of the structural encoder (rows that are identical by construction are
 * computed once: digae_layer.py:260 starts every node from ones, so early half rounds have few distinct rows): level 1 sums runs of
 * <= 64 class members with the stage backward's neighbour pull fused, the next levels sum the partial rows.  out_row (NULL: segment s
 * writes row s): the row of `out` each segment writes — a class that fits one segment writes its final row at once, only the classes
 * with more members than that leave partial rows behind the C final ones for the next level (most colours have a few members). */
int mgv_seg_sum(int H, int64_t n_seg, const int32_t* seg_ptr, const int32_t* items, const float* direct, const float* agg,
                const int32_t* nbr_ptr, const int32_t* nbr_idx, const int32_t* out_row, float* out, void* stream);

int mgv_class_pull_sum_ws_floats(int H, int64_t N, int C);
int mgv_class_pull_sum(int H, int64_t N, const float* gy_direct, const float* gy_agg, const int32_t* nbr_ptr,
                       const int32_t* nbr_idx, const int32_t* class_id, int C, float* out, float* workspace,
                       int64_t workspace_floats, void* stream);

/* ---- levelised functional sweep (dg_ae_model_aig.py:70-97 and mig/xag/xmg siblings; arch/tfmlp.py:38-46;
 * utils/dag_utils.py:91-105 is replaced by the tile tables).  T gate types ("slots"), per slot:
 *   attn_u[T][2H] = Wk^T w_attn[H:],  Wvc[T][3H][2H] = W_ih Wv,  bvc[T][3H] = W_ih bv,  bih/bhh[T][3H].
 * order/tile_* : updated nodes sorted by (level, slot) cut into <=64-node single-slot tiles;
 * level_tile_ptr_host: HOST array [num_levels+1] of tile offsets.  hf must be zero on entry
 * (num_rounds = 1: every node is updated once from h0 = 0). */
int mgv_func_sweep_fwd(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                       const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                       const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src, const float* hs,
                       float* hf, const float* attn_u, const float* Wvc, const float* bvc, const float* bih,
                       const float* bhh, void* stream);
/* backward sweep, levels in reverse.  ghf[N][H] = dL/dhf from the losses; ghs[N][H] is ADDED to;
 * scratch: dzb[N][2H], alpha[E], dsc[E] (in-CSR edge order).  WvcT[T][2H][3H]. */
int mgv_func_sweep_bwd(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                       const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                       const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src,
                       const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_slot,
                       const uint8_t* gslot, const float* hs, const float* hf, const float* attn_u,
                       const float* Wvc, const float* WvcT, const float* bvc, const float* bih, const float* bhh,
                       const float* ghf, float* ghs, float* dzb, float* alpha, float* dsc, float* d_attn_u,
                       float* dWvc, float* dbvc, float* dbih, float* dbhh, void* stream);
/* rounds r >= 2 on the fp32 kernels (dg_ae_model_aig.py:70 with num_rounds > 1): every updated gate's GRU starts from its state of
 * the previous round.  Same contract as mgv_func_sweep_round_fwd_x3 / _bwd_x3 below: hf holds the previous round's rows on entry
 * (updated rows are rewritten), gh[N][3H] = W_hh h_prev + b_hh of each node's own aggregator, h_prev[N][H], zero_bhh[T][3H] zeros;
 * backward: d_gh[N][3H] and g_hprev[N][H] = dh * z (rows of updated nodes written; the caller zeroes both), ghs is ADDED to. */
int mgv_func_sweep_round_fwd(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                             const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                             const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src, const float* hs,
                             float* hf, const float* attn_u, const float* Wvc, const float* bvc, const float* bih,
                             const float* zero_bhh, const float* gh, const float* h_prev, void* stream);
int mgv_func_sweep_round_bwd(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                             const int32_t* order, const int32_t* tile_start, const int32_t* tile_count,
                             const int32_t* tile_slot, const int32_t* in_ptr, const int32_t* in_src,
                             const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_slot,
                             const uint8_t* gslot, const float* hs, const float* hf, const float* attn_u,
                             const float* Wvc, const float* WvcT, const float* bvc, const float* bih, const float* zero_bhh,
                             const float* ghf, float* ghs, float* dzb, float* alpha, float* dsc, float* d_attn_u,
                             float* dWvc, float* dbvc, float* dbih, float* dbhh_unused, const float* gh, const float* h_prev,
                             float* d_gh, float* g_hprev, void* stream);

/* the sweep on bf16x3 split-precision MFMA (H in {32, 64}, T <= 6).  Differences from the fp32 entry points:
 * wpack_bf16[T][4][6H^2] = per slot {Wvc_hi, Wvc_lo, WvcT_hi, WvcT_lo} as bf16 in MFMA fragment order
 * (blocks (row tile, k-step) of 512 elements, lane 16q+r holds W[16 rt + r][32 ks + 8q .. +7]);
 * order_span[n_active][order_span_ints]: order_span_ints = 4: {in_ptr[v], in_ptr[v+1], out_ptr[v], out_ptr[v+1]} of v = order[i] (the CSR
 * spans in sweep order, so a tile reaches its edge lists with one load per row); order_span_ints = 32: the packed rows of
 * mgv_plan_order_rows (spans + the first in-edge sources and consumers: the lists themselves arrive with that one load).
 * Backward: no float atomics per tile.  The level kernels leave their rows' gate gradients and zbar rows in
 * `scratch` (sweep order) and one weight-gradient kernel per slot forms dWvc afterwards from the slot's tile
 * list: slot_tiles[num_tiles] = tile ids grouped by slot, slot_tile_ptr_host = HOST array [T+1] of offsets into
 * it; the small gradients (d_attn_u, dbvc, dbih, dbhh) are summed per workgroup in `scratch` too.
 * scratch_elems >= n_active * 5H + (tiles of the widest level) * T * 11H + 256 * 6H^2 floats, n_active = length of `order`
 * (the last term: per-workgroup rows of the weight-gradient kernel, summed in a fixed order: no float atomics at H = 64).
 * dWvc stays fp32 [T][3H][2H] and is ADDED to; ghs[N][H] is WRITTEN for every node (no zero fill needed). */
/* hf[v] = 0 for the nodes the sweep never updates (gslot[v] == 255; dg_ae_model_aig.py:61 zero-fills the whole state):
 * with it the caller hands the sweep an UNINITIALISED hf instead of a zero-filled one */
int mgv_sweep_zero_inactive(int H, int64_t N, const uint8_t* gslot, float* hf, void* stream);
int mgv_func_sweep_fwd_x3(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                          const int32_t* order, const int32_t* order_span, int order_span_ints, const int32_t* tile_start,
                          const int32_t* tile_count, const int32_t* tile_slot, const int32_t* in_ptr,
                          const int32_t* in_src, const float* hs, float* hf, const float* attn_u,
                          const void* wpack_bf16, const float* bvc, const float* bih, const float* bhh, void* stream);
int mgv_func_sweep_bwd_x3(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                          const int32_t* order, const int32_t* order_span, int order_span_ints, int64_t n_active,
                          const int32_t* tile_start, const int32_t* tile_count, const int32_t* tile_slot,
                          const int32_t* slot_tiles, const int32_t* slot_tile_ptr_host, const int32_t* in_ptr,
                          const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                          const int32_t* out_slot, const uint8_t* gslot, const float* hs, const float* hf,
                          const float* attn_u, const void* wpack_bf16, const float* bvc, const float* bih,
                          const float* bhh, const float* ghf, float* ghs, float* dzb, float* alpha, float* dsc,
                          float* d_attn_u, float* dWvc, float* dbvc, float* dbih, float* dbhh, float* scratch,
                          int64_t scratch_elems,
                          int skip_inactive_longer_than /* > 0: never-updated nodes with more consumers are left to mgv_sweep_pull_heavy */,
                          /* updated gates with more than skip_active_longer_than consumers (an inverter of a clock-like input), ordered by
                           * (level, id): nodes[K], node_seg_ptr[K+1], segment bounds, per-level ranges of nodes and segments as HOST arrays
                           * [num_levels + 1] (GraphPlan.heavy_segments(True, active_by_level=True)); heavy_ws: (K + segments) * 2H floats.
                           * Their pulls run per level by whole workgroups in front of the level's kernel.  0 / NULLs: none. */
                          int heavy_active_n, const int32_t* heavy_nodes, const int32_t* heavy_node_seg_ptr, const int32_t* heavy_seg_e0,
                          const int32_t* heavy_seg_e1, const int32_t* heavy_lvl_k_ptr_host, const int32_t* heavy_lvl_seg_ptr_host,
                          float* heavy_ws, int skip_active_longer_than, void* stream);

/* Rounds >= 2 of the functional sweep (dg_ae_model_aig.py:70-97 with num_rounds > 1: every gate is updated again, its GRU starting
 * from the node's state of the previous round; bf16x3, H in {32, 64}).  Same arguments as the two entries above plus
 *   gh[N][3H]     W_hh h_prev + b_hh of each node's own aggregator (r, z, n blocks), formed by the caller with mgv_linear_*;
 *   h_prev[N][H]  the previous round's states; `hf` must hold a copy of them on entry (rows of never-updated nodes stay);
 *   zero_bhh      [T][3H] zeros (b_hh is inside gh);
 * backward: d_gh[N][3H] (rows of updated nodes written) and g_hprev[N][H] = dh * z (the caller zeroes it: other rows are not written);
 * the dbhh accumulator receives nothing meaningful (its gradient comes from the caller's linear kernels). */
int mgv_func_sweep_round_fwd_x3(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                          const int32_t* order, const int32_t* order_span, int order_span_ints, const int32_t* tile_start,
                          const int32_t* tile_count, const int32_t* tile_slot, const int32_t* in_ptr,
                          const int32_t* in_src, const float* hs, float* hf, const float* attn_u,
                          const void* wpack_bf16, const float* bvc, const float* bih, const float* zero_bhh,
                          const float* gh, const float* h_prev, void* stream);
int mgv_func_sweep_round_bwd_x3(int H, int64_t N, int T, int num_levels, const int32_t* level_tile_ptr_host,
                          const int32_t* order, const int32_t* order_span, int order_span_ints, int64_t n_active,
                          const int32_t* tile_start, const int32_t* tile_count, const int32_t* tile_slot,
                          const int32_t* slot_tiles, const int32_t* slot_tile_ptr_host, const int32_t* in_ptr,
                          const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                          const int32_t* out_slot, const uint8_t* gslot, const float* hs, const float* hf,
                          const float* attn_u, const void* wpack_bf16, const float* bvc, const float* bih,
                          const float* bhh, const float* ghf, float* ghs, float* dzb, float* alpha, float* dsc,
                          float* d_attn_u, float* dWvc, float* dbvc, float* dbih, float* dbhh, float* scratch,
                          int64_t scratch_elems,
                          int skip_inactive_longer_than /* > 0: never-updated nodes with more consumers are left to mgv_sweep_pull_heavy */,
                          /* updated gates with more than skip_active_longer_than consumers (an inverter of a clock-like input), ordered by
                           * (level, id): nodes[K], node_seg_ptr[K+1], segment bounds, per-level ranges of nodes and segments as HOST arrays
                           * [num_levels + 1] (GraphPlan.heavy_segments(True, active_by_level=True)); heavy_ws: (K + segments) * 2H floats.
                           * Their pulls run per level by whole workgroups in front of the level's kernel.  0 / NULLs: none. */
                          int heavy_active_n, const int32_t* heavy_nodes, const int32_t* heavy_node_seg_ptr, const int32_t* heavy_seg_e0,
                          const int32_t* heavy_seg_e1, const int32_t* heavy_lvl_k_ptr_host, const int32_t* heavy_lvl_seg_ptr_host,
                          float* heavy_ws, int skip_active_longer_than,
                          const float* gh, const float* h_prev, float* d_gh, float* g_hprev, void* stream);
/* ---- stand-alone TFMlpAggr (arch/tfmlp.py:31-46: an edge-list call outside the levelised sweep).  Attention pooling over a CSR by
 * destination: zbar[i][W] = sum_j alpha_ij x[j], alpha = softmax over i's sources of u . x[j] (PyG softmax: exp(s - max) / (sum + 1e-16));
 * the module's message is W_v zbar + b_v [deg > 0] (mgv_linear_*).  W = row width (2 * dim_hidden) in {32, 64, 128};
 * mstat / inv [N]: the softmax statistics the backward re-uses.  Backward: dx [N][W] and du [W] are ADDED to with float atomics
 * (the caller zeroes them; this entry is not on the train step). */
int mgv_attn_pool_fwd(int W, int64_t N, const int32_t* in_ptr, const int32_t* in_src, const float* x, const float* u,
                      float* zbar, float* mstat, float* inv, void* stream);
int mgv_attn_pool_bwd(int W, int64_t N, const int32_t* in_ptr, const int32_t* in_src, const float* x, const float* u,
                      const float* zbar, const float* mstat, const float* inv, const float* dzbar, float* dx, float* du,
                      void* stream);

/* ghs rows of the heavy never-updated nodes (primary inputs driving thousands of gates): consumer lists in segments, one workgroup
 * each (GraphPlan.heavy_segments(reverse=True, inactive_only=True)); partial_ws: S * H floats */
int mgv_sweep_pull_heavy(int H, int K, const int32_t* nodes, const int32_t* node_seg_ptr, int S, const int32_t* seg_e0,
                         const int32_t* seg_e1, const int32_t* out_dst, const int32_t* out_slot, const uint8_t* gslot,
                         const float* alpha, const float* dsc, const float* dzb, const float* attn_u, float* partial_ws,
                         float* ghs, void* stream);

/* ---- the levelised sweep as ONE persistent kernel per direction (csrc/sweep_persist_x3.hip; replaces the reference's Python level
 * loop dg_ae_model_aig.py:70-97 and arch/tfmlp.py:38-46 like mgv_func_sweep_*_x3, which launch once per level).  H = 64, round 1.
 * One 8-wave workgroup per CU, each dedicated to one aggregator slot with that slot's Wvc pack resident in LDS; levels are
 * separated by an XCD-hierarchical grid barrier (csrc/mgv_gridbar.h); rows other workgroups read are stored write-through; the
 * backward accumulates dWvc in registers across all levels (no per-node rows for a deferred weight-gradient pass).
 *   key_tile_ptr[num_levels*T + 1]  DEVICE: tile range of every (level, slot) key (tiles are sorted by level, then slot)
 *   wg_begin_host[T + 1]            HOST: workgroups [wg_begin[g], wg_begin[g+1]) serve slot g; wg_begin[T] = grid <= CU count
 *   slot_tile_ptr_host[T + 1]       HOST: tiles per slot (a slot with tiles must own a workgroup)
 *   sync_ws                         DEVICE: mgv_sweep_persist_sync_bytes() bytes, zeroed by the launcher before every launch
 *   sticky_status                   DEVICE: one uint32 the CALLER zeroes once; set (never cleared) when a barrier spin gave up
 * Returns MGV_EUNSUPPORTED (use the per-level launchers) for H != 64, N*2H*4 >= 4 GB, or a grid beyond the CU count.
 * mgv_sweep_persist_status copies the sticky word back (synchronises the stream): MGV_OK, or 1000 + the give-up code. */
int mgv_sweep_persist_sync_bytes(void);                                                               /* a size, not a status */
int mgv_sweep_persist_max_grid(void);                                                                 /* CU count of the current device */
int mgv_func_sweep_fwd_persist_x3(int H, int64_t N, int T, int num_levels, const int32_t* key_tile_ptr,
                                  const int32_t* wg_begin_host, const int32_t* slot_tile_ptr_host, const int32_t* order,
                                  const int32_t* order_span, const int32_t* tile_start, const int32_t* tile_count,
                                  const int32_t* in_ptr, const int32_t* in_src, const float* hs, float* hf,
                                  const float* attn_u, const void* wpack_bf16, const float* bvc, const float* bih,
                                  const float* bhh, void* sync_ws, void* sticky_status, void* stream);
/* backward: wg_slab >= mgv_sweep_persist_slab_floats(H, grid) floats (one gradient partial row per workgroup, summed per slot in
 * workgroup order: deterministic); d_attn_u / dWvc / dbvc / dbih / dbhh are ADDED to; ghs is WRITTEN for every node (the pull of the
 * never-updated nodes follows as in mgv_func_sweep_bwd_x3; skip_inactive_longer_than as there).  The plan must hold no UPDATED gate
 * with more than 64 consumers (GraphPlan.heavy_segments(True, active_by_level=True) is None): such batches use mgv_func_sweep_bwd_x3. */
int mgv_sweep_persist_slab_floats(int H, int grid);                                                   /* a size, not a status */
int mgv_func_sweep_bwd_persist_x3(int H, int64_t N, int T, int num_levels, const int32_t* key_tile_ptr,
                                  const int32_t* wg_begin_host, const int32_t* slot_tile_ptr_host, const int32_t* order,
                                  const int32_t* order_span, const int32_t* tile_start, const int32_t* tile_count,
                                  const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                  const int32_t* out_dst, const int32_t* out_slot, const uint8_t* gslot, const float* hs,
                                  const float* hf, const float* attn_u, const void* wpack_bf16, const float* bvc,
                                  const float* bih, const float* bhh, const float* ghf, float* ghs, float* dzb,
                                  float* alpha, float* dsc, float* d_attn_u, float* dWvc, float* dbvc, float* dbih,
                                  float* dbhh, float* wg_slab, int64_t wg_slab_floats, int skip_inactive_longer_than,
                                  void* sync_ws, void* sticky_status, void* stream);
int mgv_sweep_persist_status(const void* sticky_status, void* stream);


/* ---- inner-product decoder and reconstruction loss (digae_layer.py:26-29, dg_ae_model_aig.py:108-130).
 * s, t: row pointers with common row stride ld (the two halves of hs_decompose's output);
 * edge lists are int64 like the reference's edge_index rows. */
int mgv_edge_dot_fwd(int H, int64_t E, const float* s, const float* t, int ld, const int64_t* src,
                     const int64_t* dst, int sigmoid, float* out, void* stream);
int mgv_edge_dot_bwd(int H, int64_t E, const float* s, const float* t, int ld, const int64_t* src,
                     const int64_t* dst, int sigmoid, const float* gout, float* ds, float* dt, void* stream);
/* sums[0] += sum_pos -log(sigma+1e-15), sums[1] += sum_neg -log(1-sigma+1e-15); counts += {TP,FP,TN,FN}
 * (trainer.py:240-244); pred_bin[Epos+Eneg] optional */
int mgv_recon_loss_fwd(int H, const float* s, const float* t, int ld, const int64_t* pos_src, const int64_t* pos_dst,
                       int64_t Epos, const int64_t* neg_src, const int64_t* neg_dst, int64_t Eneg,
                       double* sums, uint64_t* counts, int32_t* pred_bin, double* workspace, int64_t workspace_doubles,
                       void* stream);
/* ds/dt += dL/ds, dL/dt for loss = sums[0]/Epos + sums[1]/Eneg scaled by the DEVICE scalar *gscale.
 * When the positive edges are the batch graph's own edges pass its two int32 CSRs (pos_out_* by
 * source, pos_in_* by destination): the positive half then runs as gathers without atomics;
 * NULL CSRs = arbitrary positive list, float atomics (one whole row per wave-instruction). */
int mgv_recon_loss_bwd(int H, int64_t N, const float* s, const float* t, int ld, const int64_t* pos_src, const int64_t* pos_dst,
                       int64_t Epos, const int32_t* pos_out_ptr, const int32_t* pos_out_dst, const int32_t* pos_in_ptr,
                       const int32_t* pos_in_src, const int64_t* neg_src, const int64_t* neg_dst, int64_t Eneg,
                       const float* gscale, float* ds, float* dt, void* stream);
/* the same gradient when BOTH edge sets come as CSR pairs (by source and by destination; the negatives bucketed by
 * mgv_neg_bucket): ds/dt are WRITTEN, every row once, no atomics and no zero fill */
int mgv_recon_loss_bwd_csr(int H, int64_t N, const float* s, const float* t, int ld, const int32_t* pos_out_ptr,
                           const int32_t* pos_out_dst, const int32_t* pos_in_ptr, const int32_t* pos_in_src, int64_t Epos,
                           const int32_t* neg_out_ptr, const int32_t* neg_out_dst, const int32_t* neg_in_ptr,
                           const int32_t* neg_in_src, int64_t Eneg, const float* gscale, float* ds, float* dt,
                           int skip_pos_longer_than /* > 0: positive lists longer than this are left to mgv_recon_heavy_lists */, void* stream);
/* the positive lists mgv_recon_loss_bwd_csr skipped (nodes with thousands of consumers / producers): cut into segments
 * (nodes[K], node_seg_ptr[K+1], seg_node/seg_e0/seg_e1[S]: GraphPlan.heavy_segments), one workgroup per segment, partials
 * [S][H] in partial_ws, added into out (= ds for which 0 with list = pos_out_dst, = dt for which 1 with list = pos_in_src) */
int mgv_recon_heavy_lists(int H, const float* s, const float* t, int ld, int64_t Epos, const float* gscale, int K,
                          const int32_t* nodes, const int32_t* node_seg_ptr, int S, const int32_t* seg_node, const int32_t* seg_e0,
                          const int32_t* seg_e1, const int32_t* list, int which, float* partial_ws, float* out, void* stream);
/* negative sampling of the reconstruction loss (dg_ae_model_aig.py:115-119, torch_geometric negative_sampling): E pairs
 * uniform over {(u, v): u != v, (u, v) not an edge of the CSR}, from a counter-based generator (seed); cnt_out/cnt_in
 * [N] (zeroed by the caller) receive the pairs' per-source / per-destination counts, rank_out/rank_in [E] each pair's place inside
 * its source's / destination's bucket (the count it found).  mgv_neg_bucket then buckets the pairs without atomics: out_ptr/in_ptr =
 * exclusive scans of the counts ([N+1]); outputs the pairs grouped by source (srt_src, srt_dst: int64 like edge_index rows),
 * out_dst[E] and in_src[E] (int32 CSR payloads; order inside a bucket = thread arrival: mgv_sort_lists_i32 fixes it). */
int mgv_neg_sample(int64_t N, int64_t E, uint64_t seed, const int32_t* pos_out_ptr, const int32_t* pos_out_dst,
                   int64_t* neg_src, int64_t* neg_dst, int32_t* cnt_out, int32_t* cnt_in, int32_t* rank_out, int32_t* rank_in,
                   void* stream);
int mgv_neg_bucket(int64_t E, const int64_t* neg_src, const int64_t* neg_dst, const int32_t* out_ptr, const int32_t* in_ptr,
                   const int32_t* rank_out, const int32_t* rank_in, int64_t* srt_src, int64_t* srt_dst, int32_t* out_dst, int32_t* in_src,
                   void* stream);

/* ---- functional-similarity loss (trainer.py:158-163, utils/utils.py:32-36): dis = 1 - cos(hf[a], hf[b]),
 * L1 between the z-normalised dis and z-normalised tt.  ws[8] doubles (zeroed by the caller):
 * 0 sum dis, 1 sum dis^2, 2 sum tt, 3 sum tt^2, 4 sum |zd-zt| (loss = ws[4]/P), 5-6 backward sums. */
int mgv_func_loss_fwd(int H, int64_t P, const float* hf, const int64_t* pair_a, const int64_t* pair_b,
                      const float* tt, float eps, float* dis, double* ws, double* workspace, int64_t workspace_doubles,
                      void* stream);
/* the same gradient without atomics and without a zero-filled output: every node PULLS over the pairs it belongs to, given
 * the pair lists grouped by first member (a_ptr[N+1], a_pair[P] = pair ids) and by second member (b_ptr, b_pair) — e.g. from
 * mgv_plan_csr over (pair_a, pair_b) with its edge-id outputs; dhf[N][H] is WRITTEN for every node; bit-reproducible.
 * add (nullable, [N][H]): a gradient the same rows receive from another consumer of hf (the readout, trainer.py:155-156), summed
 * into dhf on the way out instead of by a separate N x H add */
int mgv_func_loss_bwd_csr(int H, int64_t N, int64_t P, const float* hf, const int64_t* pair_a, const int64_t* pair_b, const float* tt_sim,
                          const float* dis, float eps, const double* workspace8, const float* grad_loss, const int32_t* a_ptr,
                          const int32_t* a_pair, const int32_t* b_ptr, const int32_t* b_pair, const float* add, float* dhf, void* stream);
int mgv_func_loss_bwd(int H, int64_t P, const float* hf, const int64_t* pair_a, const int64_t* pair_b,
                      const float* tt, const float* dis, float eps, const double* ws, const float* gscale,
                      float* dhf, void* stream);

/* ---- reparameterisation sampler + KL (digvae_model.py:134-142, trainer.py:146-147):
 * z = mu + exp(logstd) * eps; eps given, or NULL = drawn from the counter-based generator (seed) and
 * returned in eps_out; klsum += sum(1 + 2 logstd - mu^2 - exp(logstd)^2) */
int mgv_reparam_fwd(int64_t n, const float* mu, const float* logstd, const float* eps, uint64_t seed,
                    float* eps_out, float* z, double* klsum, void* stream);
/* dmu = gz + *gkl*klcoef*(-2mu); dlogstd = gz*eps*exp(logstd) + *gkl*klcoef*(2-2exp(2logstd)); gz/gkl may be NULL */
int mgv_reparam_bwd(int64_t n, const float* mu, const float* logstd, const float* eps, const float* gz,
                    const float* gkl, float klcoef, float* dmu, float* dlogstd, void* stream);
/* counts += {TP,FP,TN,FN} of pred_bin vs gt_bin (trainer.py:240-244) */
int mgv_confusion(int64_t n, const int32_t* pred_bin, const int32_t* gt_bin, uint64_t* counts, void* stream);

/* ---- readout MLP pieces (arch/mlp.py:27-47 = Linear, BatchNorm1d, ReLU, Dropout; dg_ae_model_aig.py:102-106)
 * colstats: sums[c] += sum_i Y[i][c], sums[C+c] += sum_i Y[i][c]^2 (BatchNorm batch statistics) */
/* Small sums (column statistics, loss sums, head gradients) are deterministic: every workgroup leaves its partials in its own row of
 * `workspace` (>= mgv_sum_workspace_doubles() doubles, one buffer per stream in flight) and a second launch adds the rows in a fixed
 * order — no floating-point atomics, bit-identical from call to call. */
int mgv_sum_workspace_doubles(void);
int mgv_colstats(int64_t N, int C, const float* Y, int ld, double* sums, double* workspace, int64_t workspace_doubles, void* stream);
/* A = dropout_p(relu(gamma*(Y-mean)*invstd+beta)); dropout mask from a counter-based hash of (seed, element) */
int mgv_bn_act_fwd(int64_t N, int C, const float* Y, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, float p_drop, uint64_t seed, float* A, void* stream);
/* dZ = dA * mask * [bn_out > 0]; sums[c] += sum dZ (= dbeta), sums[C+c] += sum dZ*xhat (= dgamma) */
int mgv_bn_act_bwd(int64_t N, int C, const float* Y, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, float p_drop, uint64_t seed, const float* dA, float* dZ, double* sums, double* workspace,
                   int64_t workspace_doubles, void* stream);
/* dY = gamma*invstd*(dZ - [batch_stats](sums[c]/N + xhat*sums[C+c]/N)) */
int mgv_bn_bwd_apply(int64_t N, int C, const float* Y, const float* mean, const float* invstd, const float* gamma,
                     const float* dZ, const double* sums, int batch_stats, float* dY, void* stream);
/* prob = A w + b (mlp.py:43, last Linear), clamped to [0,1] when clamp01 != 0 (dg_ae_model_aig.py:105) */
int mgv_readout_head_fwd(int64_t N, int C, const float* A, const float* w, const float* b, int clamp01, float* prob, void* stream);
/* given dprob[N]: dA = dy w, dw += sum dy A, db += sum dy with dy = dprob * [clamp inactive] */
int mgv_readout_head_bwd(int64_t N, int C, const float* A, const float* w, const float* b, int clamp01, const float* dprob,
                         float* dA, float* dw, float* db, double* workspace, int64_t workspace_doubles, void* stream);
/* nn.L1Loss, reduction mean (trainer.py:71,156): sum += sum |x - target|;  dx = *gscale/n * sign(x - target) */
int mgv_l1_loss_fwd(int64_t n, const float* x, const float* target, double* sum, double* workspace, int64_t workspace_doubles,
                    void* stream);
int mgv_l1_loss_bwd(int64_t n, const float* x, const float* target, const float* gscale, float* dx, void* stream);

/* ---- Adam on a flat fp32 buffer (torch.optim.Adam as constructed at trainer.py:73); grad is multiplied
 * by grad_scale first (1/world_size after an all-reduce sum) */
int mgv_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                  float beta1, float beta2, float eps, float weight_decay, float grad_scale, int64_t step, void* stream);

/* ---- on-device batch builder: what the reference recomputes in Python inside every forward —
 * torch.stack([ei[1], ei[0]]) (digae_layer.py:264), the per-level boolean masks (dg_ae_model_aig.py:72-75), the per-node edge
 * scans of `subgraph` (utils/dag_utils.py:91-105) — and at load time the levelisation rounds of top_sort
 * (utils/dag_utils.py:10-37, via return_order_info :80-88).  All index arrays int32; results equal a STABLE sort of the edges by
 * destination / source and of the nodes by (level, slot).  `status` / `done` / `maxlevel` are small DEVICE words written
 * asynchronously (0 = fine): the caller reads them once, after the calls it batches. */
int mgv_scan_exclusive_i32(int64_t n, const int32_t* in, int32_t* out, int32_t* scratch, void* stream);   /* out[n] = total; scratch: n/2048 + 2 */
int mgv_plan_csr_scratch_ints(int64_t N, int64_t E);                                                  /* a size, not a status */
int mgv_plan_csr(int64_t N, int64_t E, const int64_t* src, const int64_t* dst, int32_t* in_ptr, int32_t* in_src, int32_t* in_dst,
                 int32_t* out_ptr, int32_t* out_dst, int32_t* out_slot, int32_t* in_eid /* NULL or [E]: edge id per in-CSR slot */,
                 int32_t* out_eid /* NULL or [E] */, int32_t* scratch, int64_t scratch_ints, int32_t* status, void* stream);
/* every list vals[ptr[n] .. ptr[n+1]) ascending, in place; scratch: N + 1 + E ints.  Applied to the lists mgv_neg_bucket fills in
 * thread-arrival order: their order then no longer depends on it (equal values are interchangeable) */
int mgv_sort_lists_i32(int64_t N, int64_t E, const int32_t* ptr, int32_t* vals, int32_t* scratch, int64_t scratch_ints, void* stream);
/* ASAP levels by frontier relaxation over the out-CSR; `rounds` level steps are enqueued; done[0] == N afterwards iff complete.
 * scratch: 3 N + rounds + 2 ints */
int mgv_plan_levels(int64_t N, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_dst, int32_t* level, int rounds,
                    int32_t* scratch, int64_t scratch_ints, int32_t* done, void* stream);
/* gate id -> aggregator slot (HOST table of 256 bytes, 255 = none), levels to int32, sort key level*T+slot (-1: never updated) */
int mgv_plan_keys(int64_t N, int T, const float* gate, const int64_t* level64, const uint8_t* slot_of_gate_host256, uint8_t* gslot,
                  int32_t* level32, int32_t* key, int32_t* maxlevel, void* stream);
int mgv_plan_check_levels(int64_t E, const int32_t* in_src, const int32_t* in_dst, const uint8_t* gslot, const int32_t* level, int32_t* err,
                          void* stream);
int mgv_count_sort_scratch_ints(int64_t n, int K);                                                    /* a size, not a status */
int mgv_count_sort_i32(int64_t n, const int32_t* key, int K, int32_t* order, int32_t* key_start, int32_t* scratch, int64_t scratch_ints,
                       void* stream);
int mgv_plan_tile_counts(int K, const int32_t* key_start, int32_t* ntile, int32_t* tile_first, int32_t* scan_scratch, void* stream);
int mgv_plan_tiles(int K, int T, int L, int64_t n_active, const int32_t* key_start, const int32_t* tile_first, const int32_t* order,
                   const int32_t* in_ptr, const int32_t* out_ptr, int32_t* tile_start, int32_t* tile_count, int32_t* tile_slot,
                   int32_t* order_span, int32_t* level_tile_ptr, void* stream);
/* packed sweep rows, 32 ints per updated node in sweep order: spans, first 4 in-edge sources, first 8 consumers (node, in-CSR slot) and
 * their gate slots (plan_build.hip k_order_rows); handed to the level sweeps as order_span with order_span_ints = 32 */
int mgv_plan_order_rows(int64_t n_active, const int32_t* order, const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                        const int32_t* out_dst, const int32_t* out_slot, const uint8_t* gslot, int32_t* rows, void* stream);
int mgv_plan_pairs(int64_t N, const int32_t* in_ptr, const uint8_t* xcls, int32_t* present, int32_t* rank, int32_t* scan_scratch, int32_t* cid,
                   int32_t* cls_deg, uint8_t* cls_x, int32_t* status, void* stream);

/* ---- colour refinement for the quotient stages of the structural encoder (GraphPlan.quotient; digae_layer.py:260: every node starts
 * from ones, so after a half round a node's row depends on (feature class, previous colour, multiset of neighbour colours) only).
 * keys: a 63-bit grouping key per node (f = int64 [3][fstride] random values per previous colour; sums over the list: order
 * independent).  check: exact comparison of every node with its group's representative rep[cid[i]] — class, previous colour, degree,
 * neighbour-colour multiset; flags[0] = 1 on any difference, flags[1] = nodes with lists beyond 48 entries, left to the caller. */
int mgv_colour_keys(int64_t N, const int32_t* nbr_ptr, const int32_t* nbr_idx, const int32_t* prev, const int64_t* f, int64_t fstride,
                    const uint8_t* xcls, int key_bits, int64_t* key, void* stream);
int mgv_colour_check(int64_t N, const int32_t* nbr_ptr, const int32_t* nbr_idx, const int32_t* prev, const uint8_t* xcls, const int32_t* cid,
                     const int32_t* rep, int32_t* flags, void* stream);
/* a refinement stage's tables on the device (csrc/plan_build.hip, GraphPlan._quotient_dev): what GraphPlan.quotient composes from torch
 * sorts / scans / gathers for CPU plans (the reference has no counterpart: it runs every half round on all N rows, digae_layer.py:257-277).
 * sort_pairs: stable radix sort of n keys (key_bytes 4 or 8, unsigned, bits [0, end_bit)), order[] = the sorting permutation (int32). */
int mgv_sort_pairs_temp_ints(int key_bytes, int64_t n);                                             /* a size in 4-byte units (-1: error) */
int mgv_sort_pairs(int key_bytes, int64_t n, const void* keys_in, void* keys_out, int32_t* order, int end_bit, void* temp,
                   int64_t temp_ints, void* stream);
/* runs of equal sorted keys -> colours: cid[N], starts[C + 1] (buffer of N + 1), rep[C] (buffer of N), n_colours[0] = C;
 * scratch_ints >= 2 N + N / 2048 + 66 */
int mgv_colour_groups(int64_t N, const int64_t* skey, const int32_t* by_colour, int32_t* cid, int32_t* starts, int32_t* rep, int32_t* n_colours,
                      int32_t* scratch, int64_t scratch_ints, void* stream);
/* representatives' rows: rptr[C + 1] (rptr[C] = list entries), own[C] previous colour, xrep[C] feature class, n_heavy[0];
 * scratch_ints >= C + C / 2048 + 66; then their lists in previous colours ent[] and the owning colour row[] of every entry */
int mgv_colour_rep_rows(int64_t C, const int32_t* rep, const int32_t* nbr_ptr, const int32_t* prev, const uint8_t* xcls, int heavy_row,
                        int32_t* rptr, int32_t* own, uint8_t* xrep, int32_t* n_heavy, int32_t* scratch, int64_t scratch_ints, void* stream);
int mgv_colour_rep_lists(int64_t C, const int32_t* rep, const int32_t* nbr_ptr, const int32_t* nbr_idx, const int32_t* prev, const int32_t* rptr,
                         int32_t* ent, int32_t* row, void* stream);
int mgv_sorted_key_counts(int64_t n, const int32_t* sorted_keys, int64_t K, int32_t* counts, void* stream);
/* one level of mgv_seg_sum's segment tables: scan leaves work[0..3] = {segments, members, partial rows, colours with > 1 segment},
 * fill writes seg_ptr[n_seg + 1], out_row[n_seg] (nullable) and the next level's colours (nullable pair) */
int mgv_seg_level_work_ints(int64_t G);                                                             /* a size */
int mgv_seg_level_scan(int64_t G, const int32_t* counts, int seg, int32_t* work, int64_t work_ints, void* stream);
int mgv_seg_level_fill(int64_t G, int64_t n_seg, const int32_t* gid, int seg, int base, const int32_t* work, int32_t* seg_ptr, int32_t* out_row,
                       int32_t* gid_next, int32_t* counts_next, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGVAE_HIP_H */
