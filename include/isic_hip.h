/* libisic_hip.so -- C ABI of the MI355X (gfx950) attention-MIL + patch-graph GNN
 * training path.
 *
 * The reference (rbuler/multimodal-isic) has NO FFI / plugin layer: its hot path
 * is plain torch-CPU/CUDA Python.  Each entry point below therefore names the
 * reference ARITHMETIC it replaces (file:line into the reference tree); the
 * Python classes that keep the reference's import surface
 * (multimodal-isic_amd/utils_g_mil.py, model.py, build_graphs.py ...) call these
 * through ctypes (multimodal-isic_amd/isic_hip/lib.py); INTEGRATION.md shows
 * the binding.
 *
 * Conventions (all entry points):
 *   - return 0 (ISIC_OK) or a negative ISIC_ERR_* code; never throw, never
 *     allocate/free device memory, never synchronise the device;
 *   - every pointer is a DEVICE pointer owned by the caller, 16-byte aligned,
 *     row-major, contiguous unless a leading dimension is given;
 *   - `stream` is a hipStream_t (passed as void*); work is asynchronous w.r.t.
 *     the host and re-entrant across streams; no global mutable state;
 *   - index tensors are int64 (the reference's `edge_index` / offsets dtype);
 *   - dropout is counter-based (Philox4x32-10): element i of a site keeps iff
 *     word_i(seed, stream_id) >= drop_threshold, kept values are multiplied by
 *     drop_scale; drop_threshold == 0 disables dropout.
 */
#ifndef ISIC_HIP_H
#define ISIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISIC_OK 0
#define ISIC_ERR_BAD_ARG (-1)
#define ISIC_ERR_UNSUPPORTED (-2)
#define ISIC_ERR_WORKSPACE (-3)
#define ISIC_ERR_LAUNCH (-4)

#define ISIC_ACT_NONE 0
#define ISIC_ACT_RELU 1
#define ISIC_ACT_TANH 2

/* ABI version of this header; bumped on any signature change. */
int isic_abi_version(void);
/* Name of the code object's target ("gfx950"). */
const char* isic_target_arch(void);

/* ------------------------------------------------------------------ dense (fp32 MFMA)
 * C[M,N] = act(op(A)[M,K] * op(B)[K,N] + bias[N]) + beta*C,  exact-fp32
 * v_mfma_f32_16x16x4_f32.  Replaces every nn.Linear on the path:
 * utils_g_mil.py:49-63 (feature_extractor / attention / patch_classifier),
 * 05_train_gnns.py:66,112,126-139 (input_proj, mlp layers, attention heads,
 * classifier), model.py:74-83,138-143 (radiomics_mlp, fusion_mlp) and their
 * autograd backward (dX = dY*W, dW = dY^T*X).
 * transA/transB: 0 = stored [rows, K] / [K, cols]; 1 = stored transposed. */
int isic_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                  float* C, int ldc, const float* bias, int act, float beta, void* stream);
/* The same product with a caller-owned workspace (isic_gemm_f32_workspace_bytes(...) bytes, 16-byte aligned; may be
 * NULL / 0).  Large products (>= 1e8 multiply-adds, every dimension a multiple of 4 floats, 16-byte aligned operands)
 * run on a persistent LDS-DMA-staged 256 x 128 x 32 kernel; a long reduction into a small output (the weight gradients
 * dW = dY^T X over all the nodes of a step) is split over K with the partial tiles parked in the workspace and added in
 * split order -- bit-reproducible.  Without a workspace such a product runs unsplit (or, on the small-tile kernel, with
 * fp32 atomics in arrival order). */
size_t isic_gemm_f32_workspace_bytes(int transA, int transB, int M, int N, int K);
int isic_gemm_f32_ws(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int act, float beta, void* workspace, size_t workspace_bytes,
                     void* stream);
/* ... with a row index on an operand: stored row r of A (of B) is read from A[a_rows[r]] (B[b_rows[r]]).  A step's batch of
 * graphs is multiplied straight out of the resident record store -- x[sel] W^T and dY^T x[sel] -- without materialising the
 * gathered node features (05_train_gnns.py:340-343 re-uploads them every step).  The index arrays are read in groups of 8
 * (allocate a multiple of 8 entries).  Supported for the operand that forms the row tiles of the persistent kernel
 * (A of a large product with > 128 output rows; B of a transA product with <= 128 output rows); ISIC_ERR_UNSUPPORTED
 * otherwise -- gather, then isic_gemm_f32_ws.  Both NULL: isic_gemm_f32_ws. */
int isic_gemm_f32_rows_ws(int transA, int transB, int M, int N, int K, const float* A, int lda, const int32_t* a_rows,
                          const float* B, int ldb, const int32_t* b_rows, float* C, int ldc, const float* bias, int act,
                          float beta, void* workspace, size_t workspace_bytes, void* stream);
/* ... + addend[M, N] (leading dimension ldadd; NULL: isic_gemm_f32_ws) added to the result: the second gradient path of a
 * residual connection joins in the GEMM's epilogue instead of in an elementwise pass over both
 * (05_train_gnns.py:187-199: h = h_prev + dropout(relu(LN(conv(h_prev)))) -> dh_prev = dY + dConv W). */
int isic_gemm_f32_add_ws(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, const float* bias, int act, float beta, const float* addend, int ldadd,
                         void* workspace, size_t workspace_bytes, void* stream);

/* out[n] = sum_m X[m,n] (+ beta*out): bias gradients of the layers above. */
int isic_colsum_f32(const float* X, int M, int N, int ldx, float* out, float beta, void* stream);
/* ... with a workspace (isic_colsum_f32_workspace_bytes(M, N) bytes; may be NULL / 0): the row chunks of a tall matrix
 * are then added in chunk order instead of through fp32 atomics -- bit-reproducible. */
size_t isic_colsum_f32_workspace_bytes(int M, int N);
int isic_colsum_f32_ws(const float* X, int M, int N, int ldx, float* out, float beta, void* workspace,
                       size_t workspace_bytes, void* stream);

/* y = act'(...) helpers for backward: dx = dy * (1 - t*t)  (tanh, from its output t). */
int isic_tanh_bwd_f32(const float* dy, const float* t, float* dx, int64_t n, void* stream);

/* y = dropout(relu(x)) in place on x (x already holds the pre-activation);
 * replaces nn.ReLU + nn.Dropout of utils_g_mil.py:50-52.  bwd: dx = dy * mask,
 * mask recomputed from y > 0 (kept & positive) and the same counter stream. */
int isic_relu_dropout_fwd_f32(float* x, int64_t n, uint32_t drop_threshold, float drop_scale, uint64_t seed,
                              uint64_t stream_id, void* stream);
int isic_relu_dropout_bwd_f32(const float* y, float* dy, int64_t n, float drop_scale, void* stream);
/* ... out of place: dx = dy * mask (the upstream gradient is left intact: no copy in front of an in-place pass) */
int isic_relu_dropout_bwd_out_f32(const float* y, const float* dy, float* dx, int64_t n, float drop_scale, void* stream);

/* dst[i][0..count[i]) = (accumulate ? dst[i] : 0) + src[i][0..count[i])  for nseg <= 32 segments in ONE launch.  dst, src and
 * count are HOST arrays (read during the call; the pointers in them are device pointers).  The per-head attention
 * parameters of GraphMIL (05_train_gnns.py:126-131: heads x {Linear, Linear}) are gathered into the fused [heads*A, H] operands
 * with one launch, and their gradients scattered back into the parameters' .grad with one, instead of 4 concatenations and 16
 * accumulation kernels per step. */
int isic_multi_copy_f32(int nseg, float* const* dst, const float* const* src, const int64_t* count, int accumulate,
                        void* stream);

/* ------------------------------------------------------------------ GraphMIL classifier head + loss, forward and backward
 * classifier_light of 05_train_gnns.py:136-139 (Linear(H, D) -> ReLU -> Dropout -> Linear(D, C)), softmax (:213-217) and
 * F.cross_entropy(log(p + 1e-9), y) (:344) in two launches instead of sixteen dependent ones:
 *   isic_graph_head_fwd_bwd  -> probs[B, C], loss_per_sample[B], loss_mean[1], dz[B, H] (for d loss = 1) and the blocks'
 *                               contributions to the parameter gradients in `workspace` (isic_graph_head_workspace_bytes);
 *                               `counter`: one zero uint32 (left zero); dropout as isic_relu_dropout_fwd_clk_f32 on [B, D];
 *   isic_graph_head_param_grads: dW1[D, H], db1[D], dW2[C, D], db2[C] (+= when accumulate) = grad_scale[0] (device scalar,
 *                               NULL = 1) * the contributions added in block order.  C <= 15. */
/* ISIC_OK when the fused head handles the shape (C < 16 and W1 twice + the row tiles fit 160 KB of LDS: H * D <= ~19 k,
 * e.g. H = D = 128), else ISIC_ERR_UNSUPPORTED: the caller then runs the operator chain (isic_gemm_f32 /
 * isic_softmax_rows / isic_cross_entropy). */
int isic_graph_head_supported(int H, int D, int C);
size_t isic_graph_head_workspace_bytes(int B, int H, int D, int C);
int isic_graph_head_fwd_bwd(const float* z, const float* W1, const float* b1, const float* W2, const float* b2,
                            const int64_t* labels, int B, int H, int D, int C, uint32_t drop_threshold, float drop_scale,
                            uint64_t seed, uint64_t stream_id, const uint64_t* clock, float* probs, float* loss_per_sample,
                            float* loss_mean, float* dz, void* workspace, size_t workspace_bytes, uint32_t* counter,
                            void* stream);
int isic_graph_head_param_grads(const void* workspace, int B, int H, int D, int C, const float* grad_scale, float* dW1,
                                float* db1, float* dW2, float* db2, int accumulate, void* stream);

/* ------------------------------------------------------------------ device step clock (captured train steps)
 * A train step captured into a hipGraph replays the SAME kernel arguments, but two values must change from step to
 * step: the dropout stream id (step * 1024 + site, oracle/philox.py) and Adam's step count t.  The `_clk` forms of the
 * entries that consume them take a device pointer to  clock[2] = {dropout step, optimizer steps taken}  and form
 * stream = stream_id + clock[0] * 1024  /  t = clock[1] + 1  on the device; isic_step_clock_advance is the last
 * kernel of the step.  With clock == NULL they are the plain entries.  (Reference loop being captured:
 * 05_train_gnns.py:336-346 / 01_train_mil_teacher.py:237-246.) */
int isic_step_clock_advance(uint64_t* clock, int dropout_steps, int optimizer_steps, void* stream);
int isic_relu_dropout_fwd_clk_f32(float* x, int64_t n, uint32_t drop_threshold, float drop_scale, uint64_t seed,
                                  uint64_t stream_id, const uint64_t* clock, void* stream);

/* ------------------------------------------------------------------ attention pool over variable-length bags
 * One workgroup per bag / graph; bags given CSR-style by offsets[B+1] into the
 * T = offsets[B] instances.  For head k (of `heads`):
 *     s[n,k]   = t[n, k*A:(k+1)*A] . w3[k,:] + b3[k]      (t = tanh(h W2^T + b2))
 *     att[n,k] = softmax over the bag's n of s[n,k]
 *     z[b,:]   = mean_k sum_n att[n,k] * h[n,:]           (feature-space pooling)
 * and, when W4 != NULL (teacher form, heads must be 1):
 *     P[n,:] = W4 h[n] + b4 ; patch_probs = softmax_c(P) ; bag_logits[b] = sum_n att[n] P[n] ;
 *     bag_probs = softmax(bag_logits).
 * Replaces utils_g_mil.py:72-97 (AttentionMIL_teacher), utils_g_mil.py:32-33
 * (AttentionMIL) and the multi-head pool 05_train_gnns.py:205-213.
 * Any output pointer may be NULL (skipped) except att.  max_bag >= the longest
 * bag (the per-bag scores live in LDS: max_bag*heads <= ~30k); H <= 1024, C <= 16. */
int isic_attn_pool_fwd(const float* h, const float* t, const float* w3, const float* b3, const float* W4,
                       const float* b4, const int64_t* offsets, int B, int H, int A, int heads, int C, int max_bag,
                       float* att, float* z, float* patch_logits, float* patch_probs, float* bag_logits,
                       float* bag_probs, void* stream);
/* Backward of the pool.  Inputs: saved h, t, att, patch_logits; upstream
 * d_bag_logits[B,C] (may be NULL) and d_z[B,H] (may be NULL).  Outputs:
 *   d_h[T,H]  (+= when accumulate_dh != 0): W4^T dP + att * d_z / heads
 *   d_u[T,heads*A]: gradient w.r.t. the PRE-tanh attention hidden (ds * w3 * (1-t^2))
 *   d_s[T,heads]: gradient of the raw scores, d_P[T,C]: gradient of patch logits
 * The weight gradients follow with isic_gemm_f32 / isic_colsum_f32
 * (dW4 = d_P^T h, dw3 = d_s^T t, dW2 = d_u^T h ...). */
int isic_attn_pool_bwd(const float* h, const float* t, const float* att, const float* patch_logits,
                       const float* w3, const float* W4, const int64_t* offsets, int B, int H, int A, int heads,
                       int C, int max_bag, const float* d_bag_logits, const float* d_z, float* d_h,
                       int accumulate_dh, float* d_u, float* d_s, float* d_P, void* stream);
/* ... that also takes the column sums the attention parameters' gradients need while d_u and t are in registers:
 * param_sums[B][2*heads*A + heads] = per bag, summed over the bag's rows in a fixed order:  d_u  |  d_s[., k] * t[., k*A + j]  |
 * d_s.  Summed over the bags (isic_colsum_f32) they ARE db2 (first Linear's bias), dw3 (second Linear's weight, [heads][A])
 * and db3 -- instead of a [heads x heads*A x T] product of which only the block diagonal is used, two column-sum passes
 * over d_u / d_s and a concatenation (05_train_gnns.py:126-139's autograd backward).  H <= 128, A <= 128, d_u != NULL;
 * ISIC_ERR_UNSUPPORTED otherwise.  param_sums == NULL: isic_attn_pool_bwd. */
int isic_attn_pool_bwd_sums(const float* h, const float* t, const float* att, const float* patch_logits,
                            const float* w3, const float* W4, const int64_t* offsets, int B, int H, int A, int heads,
                            int C, int max_bag, const float* d_bag_logits, const float* d_z, float* d_h,
                            int accumulate_dh, float* d_u, float* d_s, float* d_P, float* param_sums, void* stream);

/* ------------------------------------------------------------------ LayerNorm (+ReLU +dropout +residual)
 * y = dropout(relu?(LN(x) * gamma + beta)) + residual      rows of length N.
 * Replaces 05_train_gnns.py:187-199 (LayerNorm -> relu -> dropout -> +h_prev)
 * and model.py:75-82 (Linear -> LayerNorm -> ReLU -> Dropout).
 * Saves mean[M], rstd[M] for backward.  residual may be NULL. */
int isic_layernorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                       float* mean, float* rstd, int M, int N, float eps, int relu, uint32_t drop_threshold,
                       float drop_scale, uint64_t seed, uint64_t stream_id, void* stream);
/* dx (and d_residual = dy, taken by the caller) ; dgamma/dbeta are ACCUMULATED
 * (+=) into zero-initialised buffers. */
int isic_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, float* dx, float* dgamma, float* dbeta, int M, int N, int relu,
                       uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id, void* stream);
/* ... with the dropout step taken from a device step clock (see "device step clock" above; clock may be NULL) */
int isic_layernorm_fwd_clk(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                           float* mean, float* rstd, int M, int N, float eps, int relu, uint32_t drop_threshold,
                           float drop_scale, uint64_t seed, uint64_t stream_id, const uint64_t* clock, void* stream);
int isic_layernorm_bwd_clk(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                           const float* rstd, float* dx, float* dgamma, float* dbeta, int M, int N, int relu,
                           uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id,
                           const uint64_t* clock, void* stream);

/* ... with a workspace (isic_layernorm_bwd_workspace_bytes(N) bytes; may be NULL / 0): the blocks' dgamma / dbeta partial
 * rows are then added in block order instead of through fp32 atomics -- bit-reproducible. */
size_t isic_layernorm_bwd_workspace_bytes(int N);
int isic_layernorm_bwd_ws(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                          const float* rstd, float* dx, float* dgamma, float* dbeta, int M, int N, int relu,
                          uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id,
                          const uint64_t* clock, void* workspace, size_t workspace_bytes, void* stream);

/* ... that also ACCUMULATES (+=) the column sums of dx into dxsum[N]: when the normalised tensor is A^ (h W^T) + b, they are
 * the gradient of the bias b (GCNConv's, 05_train_gnns.py:82,184-187) -- taken while dx is in registers instead of by a
 * column-sum pass over it.  N in {64, 128, 256}, 16-byte aligned, workspace required (ISIC_ERR_UNSUPPORTED / _WORKSPACE
 * otherwise); dxsum == NULL: isic_layernorm_bwd_ws. */
int isic_layernorm_bwd_dxsum_ws(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                                const float* rstd, float* dx, float* dgamma, float* dbeta, float* dxsum, int M, int N,
                                int relu, uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id,
                                const uint64_t* clock, void* workspace, size_t workspace_bytes, void* stream);

/* y = x / max(||x||_2, eps) per row and its backward: F.normalize of
 * SAGEConv(normalize=True) (05_train_gnns.py:87-88). */
int isic_l2normalize_fwd(const float* x, float* y, float* norm, int M, int N, float eps, void* stream);
int isic_l2normalize_bwd(const float* dy, const float* y, const float* norm, float* dx, int M, int N, float eps,
                         void* stream);

/* ------------------------------------------------------------------ loss
 * Per-sample cross entropy, mean over B, and its gradient scaled by
 * grad_scale/B.  mode 0: inputs are logits (01_train_mil_teacher.py:143,244);
 * mode 1: inputs are PROBABILITIES and the loss is CE(log(p + 1e-9))
 * (05_train_gnns.py:344), gradient w.r.t. p.  loss[1] receives the mean. */
int isic_cross_entropy(const float* in, const int64_t* labels, int B, int C, int mode, float grad_scale,
                       float* loss_per_sample, float* loss_mean, float* d_in, void* stream);
/* probs = softmax(logits) rows, and backward d_logits = p * (d_p - sum(d_p * p)). */
int isic_softmax_rows_fwd(const float* logits, float* probs, int M, int N, void* stream);
int isic_softmax_rows_bwd(const float* probs, const float* d_probs, float* d_logits, int M, int N, void* stream);

/* ------------------------------------------------------------------ optimizer
 * torch.optim.AdamW / Adam single-tensor semantics (01_train_mil_teacher.py:217-224,
 * 05_train_gnns.py:332-333) on one flat fp32 buffer, in torch's order of
 * operations.  The caller computes the scalar factors in double precision:
 *   step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t),
 *   AdamW: decay_factor = 1 - lr*wd, l2 = 0;   Adam: decay_factor = 1, l2 = wd.
 * grad_scale multiplies g first (1/world_size after a sum all-reduce).
 * If p_bf16 != NULL a bf16 copy of the updated parameters is written too. */
int isic_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float step_size, float beta1, float beta2,
                   float eps, float decay_factor, float l2, float bias_correction2_sqrt, float grad_scale,
                   uint16_t* p_bf16, void* stream);
/* ... with t = clock[1] + 1 read on the device: step_size = lr / (1 - beta1^t) (lr in double, one rounding to fp32 as on the
 * host), bias_correction2_sqrt = sqrt(1 - beta2^t) */
int isic_adam_step_clk(float* p, const float* g, float* m, float* v, int64_t n, double lr, float beta1, float beta2,
                       float eps, float decay_factor, float l2, float grad_scale, uint16_t* p_bf16, const uint64_t* clock,
                       void* stream);

/* ------------------------------------------------------------------ patch-graph adjacency + message passing
 * k-NN graph on node features (03_build_graphs.py:37-54, utils_g_mil.py:596-615):
 * per graph g (nodes offsets[g]..offsets[g+1]) d = (|xi|^2 + |xj|^2) - 2 xi.xj in
 * fp32 (exact-fp32 MFMA dot products), clamp >= 0, diagonal = +inf, the k smallest
 * per row in ascending order (ties: lower index first).  Writes nn_idx[T,k] as
 * LOCAL node ids (-1 past a graph's N-1 neighbours) and optionally nn_dist[T,k].
 * The caller forms edge_index = [repeat(i,k); nn_idx] exactly as the reference.
 * max_nodes >= the largest graph (<= ~2400), workspace_sqnorm: T floats. */
int isic_knn_graph(const float* x, const int64_t* offsets, int G, int D, int k, int max_nodes, int64_t total_nodes,
                   int64_t* nn_idx, float* nn_dist, float* workspace_sqnorm, void* stream);

/* Lesion-mask -> patch flags of the latent extraction path (save_latent.py:73-87): mask[B,H,W] fp32,
 * flags[B, H/patch, W/patch] = 1 iff the patch x patch pixel block holds any value > 0 (H, W multiples of patch). */
int isic_mask_patch_flags_f32(const float* mask, uint8_t* flags, int64_t B, int H, int W, int patch, void* stream);

/* Edge-attention message passing for the remaining PyG layers GraphMIL can select (05_train_gnns.py:94-106), on the
 * destination-major CSR (rowptr/col) and its transpose (rowptr_t/col_t/perm_t) of isic_gcn_csr_build.  Tensors are
 * [N, H, F] fp32 row-major; alpha/de are [nnz, H].
 *   mode 0 (GATv2Conv, 05:99-101):    logit = sum_f att[h,f] * leaky_relu(ks[src] + qd[dst]); values v == ks (= lin_l(x));
 *                                     qd = lin_r(x); CSR built in mode 'gcn' (self loops re-added).
 *   mode 1 (TransformerConv, 05:94-98): logit = scale * <qd[dst], ks[src]>, ks = key, qd = query, v = value; CSR built in
 *                                     mode 'sum' (edges as given); a node without incoming edges gets out = bias.
 * alpha = softmax of the logits over the edges INTO each destination, per head; dropout (Philox word p*H+h) on alpha;
 * out[dst,h,:] = sum alpha * v[src,h,:] (+ bias[H*F]).  fwd writes the pre-dropout alpha.
 * bwd: de[nnz,H] = d logit (workspace and output), dqd = d loss / d qd, dks = d loss / d ks (mode 0: including the
 * value path, v == ks), dv = d loss / d v (mode 1 only), datt[H,F] += d loss / d att (mode 0 only; zero it first; any
 * H, F: kept in registers across rows while H * ceil(F/64) <= 16 -- the reference class defaults, hidden 256 x 4 heads,
 * included -- one atomic per element and row beyond that). */
int isic_edge_attn_fwd(int mode, const float* ks, const float* qd, const float* v, const float* att, const int32_t* rowptr,
                       const int32_t* col, const float* bias, float* out, float* alpha, int64_t N, int H, int F,
                       float negative_slope, float scale, uint32_t drop_threshold, float drop_scale, uint64_t seed,
                       uint64_t stream_id, void* stream);
int isic_edge_attn_bwd(int mode, const float* dout, const float* ks, const float* qd, const float* v, const float* att,
                       const float* alpha, const int32_t* rowptr, const int32_t* col, const int32_t* rowptr_t,
                       const int32_t* col_t, const int32_t* perm_t, float* de, float* dqd, float* dks, float* dv, float* datt,
                       int64_t N, int H, int F, float negative_slope, float scale, uint32_t drop_threshold, float drop_scale,
                       uint64_t seed, uint64_t stream_id, void* stream);
/* FAConv (05:102-105, restated from the published layer: the reference's call passes the wrong arguments):
 * out[dst,:] = sum_e tanh(al[src] + ar[dst]) * val[e] * x[src,:] + eps * x0[dst,:] with al = <x, att_l>, ar = <x, att_r>
 * (isic_gat_scores with H = 1) and val = the GCN normalisation of a 'gcn'-mode CSR; dropout on the tanh scores (Philox
 * word p).  coef[nnz] receives the pre-dropout tanh score.  bwd: de[nnz], dar[N], dal[N] are outputs / workspace,
 * dx[N,F] = d loss / d x including the att_l / att_r paths (d att_l = dal^T x, d att_r = dar^T x and d x0 = eps * dout
 * are formed by the caller). */
int isic_fa_fwd(const float* x, const float* x0, const float* al, const float* ar, const int32_t* rowptr, const int32_t* col,
                const float* val, float* out, float* coef, int64_t N, int F, float eps, uint32_t drop_threshold,
                float drop_scale, uint64_t seed, uint64_t stream_id, void* stream);
int isic_fa_bwd(const float* dout, const float* x, const float* coef, const float* att_l, const float* att_r,
                const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* rowptr_t, const int32_t* col_t,
                const float* val_t, const int32_t* perm_t, float* de, float* dar, float* dal, float* dx, int64_t N, int F,
                uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id, void* stream);

/* Edge-wise heterophily measures of a patch graph (04_measure_heterophily.py:107-169), per directed edge
 * e = (src[e] -> dst[e]) of a batch of graphs (global node ids; every graph has nodes_per_graph nodes laid out on a
 * grid_w-wide lattice, 04:124-125):
 *   h_kl[e]        = sum_c p[src,c] * log((p[src,c] + eps) / (p[dst,c] + eps))      (:164)
 *   h_dirichlet[e] = 0.5 * || x[src] - x[dst] ||^2                                  (:165)
 *   h_spatial[e]   = Euclidean lattice distance of the two patches                  (:166)
 *   same_class[e]  = 1.0 iff dominant_class[src] == dominant_class[dst]             (:130)
 * x[T,D], probs[T,C] fp32, dominant_class[T] int32, src/dst[E] int64.  Self loops are NOT dropped (the reference
 * strips them first, :117-118): the caller filters on src != dst, which keeps the reference's edge order. */
int isic_edge_heterophily_f32(const float* x, const float* probs, const int32_t* dominant_class, const int64_t* src,
                              const int64_t* dst, int64_t num_edges, int D, int C, int grid_w, int nodes_per_graph,
                              float eps, float* h_kl, float* h_dirichlet, float* h_spatial, float* same_class,
                              void* stream);

/* CSR-by-destination with GCN symmetric normalisation (PyG GCNConv.gcn_norm as
 * called at 05_train_gnns.py:82,184-185): existing self loops are dropped, one
 * self loop (weight 1, or the dropped loop's weight) is added per node,
 * deg[i] = sum of weights into i, w^ = deg^-1/2[src] * w * deg^-1/2[dst].
 * src/dst are GLOBAL node ids over the batched graphs (n_nodes total).
 * Outputs: rowptr[n_nodes+1], col[E+n_nodes] (source ids; edges keep their
 * edge_index order within a row, self loop last), val[E+n_nodes]; and the
 * transposed structure (CSR by source) rowptr_t/col_t/val_t for the backward
 * pass.  workspace: isic_gcn_csr_workspace_bytes(n_nodes, E) bytes.
 * mode 0 = the GCN normalisation above; mode 1 = plain sum aggregation (val = w, no
 * self loops: GINConv, 05_train_gnns.py:89-93); mode 2 = mean aggregation (val =
 * w / in-degree: SAGEConv(aggr='mean'), 05_train_gnns.py:87-88).
 * perm_t (optional, [E+n_nodes]): for every slot of the transposed CSR the slot of the
 * same edge in the destination-major CSR (needed by the attention backward). */
size_t isic_gcn_csr_workspace_bytes(int64_t n_nodes, int64_t E);
int isic_gcn_csr_build(const int64_t* src, const int64_t* dst, const float* edge_weight, int64_t E, int64_t n_nodes,
                       int mode, int32_t* rowptr, int32_t* col, float* val, int32_t* rowptr_t, int32_t* col_t,
                       float* val_t, int32_t* perm_t, void* workspace, size_t workspace_bytes, void* stream);
/* A step's batched CSR (+ transpose) out of per-graph pieces built once: graphs sel[0..B) of a set in which every graph
 * has n nodes and nnz stored entries; inputs are [G, n] row pointers and [G, nnz] columns / values / perm_t with LOCAL ids
 * (row pointers relative to the graph's first entry), outputs the batch's arrays with global ids (rowptr: B*n + 1 entries).
 * One launch per optimizer step instead of re-uploading every graph (05_train_gnns.py:340-343). */
int isic_csr_batch_assemble(const int64_t* sel, int B, int n, int nnz, const int32_t* rowptr, const int32_t* rowptr_t,
                            const int32_t* col, const int32_t* col_t, const float* val, const float* val_t,
                            const int32_t* perm_t, int32_t* rowptr_out, int32_t* rowptr_t_out, int32_t* col_out,
                            int32_t* col_t_out, float* val_out, float* val_t_out, int32_t* perm_t_out, int32_t* row_index_out,
                            void* stream);
/* out[i,:] = alpha * sum_{e in row i} val[e] * x[col[e],:] (+ bias) (+ addend_scale*addend[i,:])
 * -- the neighbour gather / segmented sum of GCNConv.propagate
 * (05_train_gnns.py:184-185); GCN2Conv's (1-alpha) A^ x + alpha x_0 with addend.
 * With (rowptr_t, col_t, val_t) it is the backward d_x = A^T d_out. */
int isic_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, const float* bias,
                      float* out, int64_t n_rows, int F, float alpha, const float* addend, float addend_scale,
                      void* stream);
/* Graph attention (PyG GATConv(heads=H, concat=True, dropout=p) as called at
 * 05_train_gnns.py:83-86) on the GCN-mode CSR (self loops re-added; `val` unused):
 *   al[n,h] = <x'[n,h,:], att_src[h,:]>, ar[n,h] = <x'[n,h,:], att_dst[h,:]>        (isic_gat_scores)
 *   alpha = softmax over the edges into dst of leaky_relu(al[src] + ar[dst], slope), dropout on
 *   alpha (counter-based, element index = slot*H + h), out[dst,h,:] = sum alpha x'[src,h,:] + bias.
 * x' is [N, H*F]; alpha[nnz, H] (pre-dropout) is saved for the backward pass, which returns
 * d x' (aggregation + score paths), d al, d ar and uses de[nnz,H] as scratch; the att_src /
 * att_dst gradients follow as dal^T x' / dar^T x' per head (isic_gemm_f32). */
int isic_gat_scores(const float* xp, const float* att_src, const float* att_dst, float* al, float* ar, int64_t N, int H,
                    int F, void* stream);
int isic_gat_fwd(const float* xp, const float* al, const float* ar, const int32_t* rowptr, const int32_t* col,
                 const float* bias, float* out, float* alpha, int64_t N, int H, int F, float negative_slope,
                 uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id, void* stream);
int isic_gat_bwd(const float* dout, const float* xp, const float* alpha, const float* al, const float* ar,
                 const float* att_src, const float* att_dst, const int32_t* rowptr, const int32_t* col,
                 const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t, float* de, float* dar, float* dal,
                 float* dxp, int64_t N, int H, int F, float negative_slope, uint32_t drop_threshold, float drop_scale,
                 uint64_t seed, uint64_t stream_id, void* stream);

/* ------------------------------------------------------------------ patch encoder (bf16 MFMA, NHWC)
 * The reference's encoder is an un-vendored ConvMAE run through torch
 * (save_latent.py:42-60); BASELINE.json configs[1] names ResNet-18 (layer table:
 * SURVEY.md 8d).  Activations are NHWC bf16, weights are [Cout][Kh][Kw][Cin]
 * (= torch channels_last memory of an OIHW parameter), accumulation fp32.
 *
 * isic_conv2d_igemm_bf16: implicit-GEMM convolution, forward AND data gradient:
 *   out[n,ho,wo,co] = sum_{kh,kw,ci} in[n, (ho*up+kh-pad)/down, (wo*up+kw-pad)/down, ci] * w[co,kh,kw,ci]
 *                     (+ addend[n,ho,wo,co], added in fp32 before the single rounding to bf16)
 *   (taps whose source pixel is fractional or outside the image contribute zero; a down = 2
 *   launch is split by output parity so that only integral taps are visited)
 *   forward: up = stride, down = 1, pad = padding, w = bf16 W[co][kh][kw][ci]
 *   dgrad:   up = 1, down = stride, pad = k-1-padding, in = dY, w = W flipped+transposed [ci][kh][kw][co]
 * Optional fused BatchNorm statistics (down = 1 only): per-channel sum / sum of squares of the
 * rounded outputs are ADDED into row (block % stat_slots) of stat_sum/stat_sumsq[stat_slots][Cout] (zeroed by
 * the caller; isic_bn_finalize adds the rows in a fixed order).  The ResNet-18 kernels (3x3 stride 1 with >= 64
 * channels, stride-2 3x3, 1x1) are persistent with at most one block per CU: with stat_slots >= 256 every block owns
 * its row and the statistics are bit-reproducible; fewer rows (or the generic tile-per-block kernel) share rows
 * through fp64 atomics in arrival order.
 * Cin and Cout must be multiples of 64; down in {1,2}. */
int isic_conv2d_igemm_bf16(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int Hin, int Win, int Cin,
                           int Hout, int Wout, int Cout, int Kh, int Kw, int up, int down, int pad,
                           const uint16_t* addend, double* stat_sum, double* stat_sumsq, int stat_slots, void* stream);
/* dW[co][kh][kw][ci] (fp32) += sum_{n,ho,wo} dY[n,ho,wo,co] * X[n,ho*stride+kh-pad,wo*stride+kw-pad,ci]
 * (accumulates into the caller's zeroed or running gradient; no atomics: blocks write split-K partial tiles into the
 * workspace and a second kernel adds them in a fixed order -- every weight gradient is bit-reproducible).
 * workspace: 256-byte aligned, isic_conv2d_wgrad_workspace_bytes(...) bytes (per-pixel source-offset table rebuilt by
 * every call + the split-K partials). */
size_t isic_conv2d_wgrad_workspace_bytes(int N, int Cin, int Hout, int Wout, int Cout, int Kh, int Kw);
int isic_conv2d_wgrad_bf16(const uint16_t* x, const uint16_t* dy, float* dw, int N, int Hin, int Win, int Cin,
                           int Hout, int Wout, int Cout, int Kh, int Kw, int stride, int pad, void* workspace,
                           size_t workspace_bytes, void* stream);
/* fp32 master weights [O][Kh][Kw][I] -> bf16 forward copy (same order) and bf16
 * dgrad copy [I][Kh][Kw][O] with both taps flipped (either may be NULL). */
int isic_conv_weight_prep_bf16(const float* w_krsc, uint16_t* w_fwd, uint16_t* w_dgrad, int O, int I, int Kh, int Kw,
                               void* stream);
/* Data gradient of a ResNet DOWNSAMPLE block's two stride-2 convolutions in one launch: dx = dgrad3x3/2(dy; w) +
 * dgrad1x1/2(dy2; w2), both reading [N, Ho, Wo, Co] gradients and writing [N, H, W, C] (H = 2 Ho, pad 1 / pad 0).  The 1x1
 * convolution's taps land on the even pixels only: its K-tiles are appended to the 3x3 gradient's even-pixel parity class and
 * accumulate into the same registers -- instead of a kernel that writes three zeros per value (an [N, H, W, C] tensor) and
 * a second one that reads it back as an addend.  w = isic_conv_weight_prep_bf16's dgrad copy [C][3][3][Co], w2 = the 1x1
 * layer's [C][Co].  Sum order: the 3x3 taps, then the 1x1 term, in fp32, one rounding (the two-launch form rounds the 1x1
 * term to bf16 first). */
int isic_conv2d_dgrad_pair_bf16(const uint16_t* dy, const uint16_t* w, const uint16_t* dy2, const uint16_t* w2, uint16_t* dx,
                                int N, int Ho, int Wo, int Co, int H, int W, int C, void* stream);
/* isic_conv2d_igemm_bf16 with an addend that is MASKED ON THE FLY: out = conv + (bit ? addend : 0), bit = bit (c & 7) of
 * addend_mask[(pixel * Cout + c) / 8] -- the 1-bit ReLU mask isic_bn_apply_mask_bf16 writes.  In a ResNet BasicBlock the
 * data gradient of conv1 is joined with the gradient through the identity, which is d(out) where relu(bn2 + x) was active:
 * with this entry that gradient is never materialised (one 2-byte write and one 2-byte read per element less than
 * isic_bn_bwd_apply_mask_bf16(dres = ...) + isic_conv2d_igemm_bf16(addend = dres); the results are bit-identical: a
 * masked bf16 value is the value or zero).  3x3 / stride 1 / pad 1 layers served by the pixels-staged-once kernels
 * (64 -> 64; >= 128 channels with Cout % 128 == 0); ..._supported() says so, otherwise ISIC_ERR_UNSUPPORTED. */
size_t isic_conv2d_maskadd_supported(int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int Kh, int Kw, int up,
                                    int down, int pad);
int isic_conv2d_igemm_maskadd_bf16(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int Hin, int Win, int Cin,
                                   int Hout, int Wout, int Cout, int Kh, int Kw, int up, int down, int pad,
                                   const uint16_t* addend, const uint8_t* addend_mask, void* stream);
/* Data gradient whose output is the gradient g of an activation z = ReLU(BatchNorm(y) [+ residual]) (ResNet BasicBlock:
 * the data gradient of conv2 feeds bn1's backward, that of the next block's conv1 (+ identity addend) feeds bn2's).
 * The epilogue applies the ReLU mask (relu_mask[rows][Cout/8], bit c = channel 8g + c, as written by
 * isic_bn_apply_mask_bf16), stores out = dz = mask ? g (+ addend) : 0 and ADDS the per-channel sums BatchNorm's backward
 * needs, sum dz and sum dz * y_raw, into sum_dz / sum_dzy[stat_slots][Cout] (zeroed by the caller) -- the separate
 * reduction pass over (g, y) disappears, and for a block's last BatchNorm so does the materialised residual gradient
 * (it IS dz).  isic_bn_bwd_finalize turns the sums into dgamma / dbeta; isic_bn_bwd_apply_bf16(dz, y_raw, relu = 0)
 * finishes.  Geometry arguments as isic_conv2d_igemm_bf16; ..._supported() says whether a fused kernel exists for the
 * shape (3x3, stride 1, pad 1, Cin >= 128, Cout % 128 == 0, W <= 63) -- otherwise the caller runs the unfused sequence. */
size_t isic_conv2d_dgrad_bnbwd_supported(int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int Kh, int Kw,
                                      int up, int down, int pad);
int isic_conv2d_dgrad_bnbwd_bf16(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int Hin, int Win, int Cin,
                                 int Hout, int Wout, int Cout, int Kh, int Kw, int up, int down, int pad,
                                 const uint16_t* addend, const uint8_t* relu_mask, const uint16_t* y_raw, double* sum_dz,
                                 double* sum_dzy, int stat_slots, void* stream);
int isic_bn_bwd_finalize(const double* sum_dz, const double* sum_dzy, int nslots, int C, const float* mean,
                         const float* rstd, double* dgamma, double* dbeta, void* stream);
/* Stem: 7x7/2 pad 3, 3->64, on NHWC input with C padded to 4.  w_stem is
 * [64][7][8][4] bf16 (kw, ci zero padded) made by isic_conv_stem_pack_bf16 from the
 * fp32 [64][7][7][3] parameter; the weight gradient accumulates (atomics) into
 * fp32 [64][7][7][3]. */
int isic_conv_stem_fwd_bf16(const uint16_t* in_nhwc4, const uint16_t* w_stem, uint16_t* out, int N, int Hin, int Win,
                            int Hout, int Wout, void* stream);
/* ... with the BatchNorm statistics of the (rounded) output fused, as isic_conv2d_igemm_bf16 does:
 * stat_sum / stat_sumsq [stat_slots][64] fp64, accumulated into (zero them first). */
int isic_conv_stem_fwd_stats_bf16(const uint16_t* in_nhwc4, const uint16_t* w_stem, uint16_t* out, int N, int Hin,
                                  int Win, int Hout, int Wout, double* stat_sum, double* stat_sumsq, int stat_slots,
                                  void* stream);
/* dw[64][7][7][3] (fp32) += weight gradient of the stem.  No atomics: every persistent block leaves its partial in
 * `workspace` (isic_conv_stem_wgrad_workspace_bytes() bytes, 256-byte aligned) and a second kernel adds the partials in
 * block order -- bit-reproducible. */
size_t isic_conv_stem_wgrad_workspace_bytes(void);
int isic_conv_stem_wgrad_bf16(const uint16_t* in_nhwc4, const uint16_t* dy, float* dw, int N, int Hin, int Win,
                              int Hout, int Wout, void* workspace, size_t workspace_bytes, void* stream);
int isic_conv_stem_pack_bf16(const float* w_krsc, uint16_t* w_stem, void* stream);
/* Stem weight gradient straight from the POOLED gradient: dw += conv_wgrad(in, dY) with
 * dY = BatchNorm(+ReLU) backward (isic_bn_bwd_apply_pooled_bf16) of the 3x3/2 max-pool backward of dy_pooled, formed in
 * registers from y0 (the stem convolution output), argmax and the reduced sums dgamma / dbeta -- the full-size dY is
 * neither written nor read.  Also adds dgamma / dbeta into the fp32 gradients when those are given.  Hp, Wp: pooled size.
 * workspace: isic_conv_stem_wgrad_workspace_bytes() bytes (per-block partials, added in block order: no atomics). */
int isic_conv_stem_wgrad_bn_pooled_bf16(const uint16_t* in_nhwc4, const uint16_t* y0, const uint8_t* argmax,
                                        const uint16_t* dy_pooled, const float* mean, const float* rstd,
                                        const float* gamma, const float* scale, const float* shift, const double* dgamma,
                                        const double* dbeta, float* dw, float* dgamma_f32, float* dbeta_f32, int N,
                                        int Hin, int Win, int Hout, int Wout, int Hp, int Wp, void* workspace,
                                        size_t workspace_bytes, void* stream);
/* NCHW fp32/bf16 images (the reference's dataset layout, dataset.py:36-40) ->
 * NHWC bf16 with C padded to 4. */
int isic_nchw_to_nhwc4_bf16(const void* in, int in_is_bf16, uint16_t* out, int N, int C, int H, int W, void* stream);

/* BatchNorm2d, training mode, NHWC bf16.  stats: per-channel sum / sum of squares
 * in fp64 (zeroed by the caller), as sum[nslots][C] partial rows; finalize adds the
 * slots: scale = gamma*rstd, shift = beta - mean*scale, running stats updated with
 * `momentum` (unbiased variance). */
int isic_bn_stats_bf16(const uint16_t* x, int64_t rows, int C, double* sum, double* sumsq, void* stream);
int isic_bn_finalize(const double* sum, const double* sumsq, int nslots, int64_t rows, int C, const float* gamma,
                     const float* beta, float eps, float momentum, float* scale, float* shift, float* mean,
                     float* rstd, float* running_mean, float* running_var, void* stream);
/* eval mode: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale */
int isic_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, int C, float* scale, float* shift, void* stream);
/* y = relu?(x*scale + shift + residual) */
int isic_bn_apply_bf16(const uint16_t* x, const float* scale, const float* shift, const uint16_t* residual,
                       uint16_t* y, int64_t rows, int C, int relu, void* stream);
/* backward in two passes.  The ReLU mask comes from y, or -- when scale/shift (the
 * forward affine) are given and there was no residual -- is recomputed as
 * x*scale+shift > 0 so that y is not read at all.  reduce: dz = dy * mask; dbeta = sum dz,
 * dgamma = sum dz*xhat (fp64, zeroed by the caller).  apply: dx = gamma*rstd *
 * (dz - dbeta/rows - xhat*dgamma/rows); d_residual = dz (optional); the fp32
 * parameter gradients get += dgamma/dbeta (optional). */
int isic_bn_bwd_reduce_bf16(const uint16_t* dy, const uint16_t* x, const uint16_t* y, const float* mean,
                            const float* rstd, int64_t rows, int C, int relu, const float* scale,
                            const float* shift, double* dgamma, double* dbeta, void* stream);
int isic_bn_bwd_apply_bf16(const uint16_t* dy, const uint16_t* x, const uint16_t* y, const float* mean,
                           const float* rstd, const float* gamma, const double* dgamma, const double* dbeta,
                           int64_t rows, int C, int relu, const float* scale, const float* shift, uint16_t* dx,
                           uint16_t* d_residual, float* dgamma_f32, float* dbeta_f32, void* stream);
/* MaxPool 3x3/2 pad 1 (first maximum wins, like torch); argmax[N,Ho,Wo,C] holds the
 * winning tap kh*3+kw for the backward pass. */
int isic_maxpool3x3s2_fwd_bf16(const uint16_t* x, uint16_t* y, uint8_t* argmax, int N, int H, int W, int C, int Ho,
                               int Wo, void* stream);
int isic_maxpool3x3s2_bwd_bf16(const uint8_t* argmax, const uint16_t* dy, uint16_t* dx, int N, int H, int W, int C,
                               int Ho, int Wo, void* stream);

/* ReLU mask as 1 bit per element (bit j of byte [row][C/8] = output channel 8*byte+j is positive): the backward
 * passes of a BatchNorm + residual + ReLU (BasicBlock.bn2, net: torchvision resnet.py BasicBlock.forward) read it
 * instead of the full output tensor -- same results as the y-based forms below, ~2 B/element less traffic per pass. */
int isic_bn_apply_mask_bf16(const uint16_t* x, const float* scale, const float* shift, const uint16_t* residual,
                            uint16_t* y, uint8_t* relu_mask, int64_t rows, int C, void* stream);
/* ... with a residual that is itself a RAW convolution output under a BatchNorm without ReLU (the 1x1 shortcut of a
 * downsample block, torchvision BasicBlock.downsample): y = relu(x * scale + shift + bf16(residual_raw * res_scale + res_shift)).
 * The shortcut's normalised tensor is never written; the inner rounding keeps the result bit-identical to
 * isic_bn_apply_bf16(residual_raw -> idn) + isic_bn_apply_mask_bf16(residual = idn). */
int isic_bn_apply_mask_res_affine_bf16(const uint16_t* x, const float* scale, const float* shift, const uint16_t* residual_raw,
                                       const float* res_scale, const float* res_shift, uint16_t* y, uint8_t* relu_mask,
                                       int64_t rows, int C, void* stream);
int isic_bn_bwd_reduce_mask_bf16(const uint16_t* dy, const uint16_t* x, const uint8_t* relu_mask, const float* mean,
                                 const float* rstd, int64_t rows, int C, double* dgamma, double* dbeta, void* stream);
int isic_bn_bwd_apply_mask_bf16(const uint16_t* dy, const uint16_t* x, const uint8_t* relu_mask, const float* mean,
                                const float* rstd, const float* gamma, const double* dgamma, const double* dbeta,
                                int64_t rows, int C, uint16_t* dx, uint16_t* d_residual, float* dgamma_f32,
                                float* dbeta_f32, void* stream);

/* Stem fusions (the 112x112x64 stem activation is the largest tensor of the network: 1.6 GB at 1024 images).
 * y = maxpool3x3s2(relu(x * scale + shift)): BatchNorm apply (net_utils.py / torchvision ResNet.forward: bn1, relu,
 * maxpool) without materialising the normalised tensor; same values and argmax as isic_bn_apply_bf16 followed by
 * isic_maxpool3x3s2_fwd_bf16. */
int isic_bn_relu_maxpool3x3s2_fwd_bf16(const uint16_t* x, const float* scale, const float* shift, uint16_t* y,
                                       uint8_t* argmax, int N, int H, int W, int C, int Ho, int Wo, void* stream);
/* Same, and additionally x_sel[N,Ho,Wo,C] = the RAW input value at each window's argmax (optional).  With it the
 * BatchNorm backward reduction of the stem is a pass over the pooled tensors only (a window's gradient reaches exactly
 * its argmax pixel): isic_bn_bwd_reduce_bf16(dy_pooled, x_sel, ..., rows = N*Ho*Wo, relu = 1, scale, shift) gives the
 * sums of isic_bn_bwd_reduce_pooled_bf16 without reading the 4x larger x. */
int isic_bn_relu_maxpool3x3s2_fwd_sel_bf16(const uint16_t* x, const float* scale, const float* shift, uint16_t* y,
                                           uint8_t* argmax, uint16_t* x_sel, int N, int H, int W, int C, int Ho, int Wo,
                                           void* stream);
/* BatchNorm(+ReLU, mask recomputed from x) backward whose incoming gradient is the max-pool backward of `dy_pooled`
 * through `argmax`, gathered on the fly: same results as isic_maxpool3x3s2_bwd_bf16 followed by
 * isic_bn_bwd_reduce_bf16 / isic_bn_bwd_apply_bf16 (relu = 1, scale/shift given), without the full-size gradient. */
int isic_bn_bwd_reduce_pooled_bf16(const uint8_t* argmax, const uint16_t* dy_pooled, const uint16_t* x, const float* mean,
                                   const float* rstd, int N, int H, int W, int C, int Ho, int Wo, const float* scale,
                                   const float* shift, double* dgamma, double* dbeta, void* stream);
int isic_bn_bwd_apply_pooled_bf16(const uint8_t* argmax, const uint16_t* dy_pooled, const uint16_t* x, const float* mean,
                                  const float* rstd, const float* gamma, const double* dgamma, const double* dbeta, int N,
                                  int H, int W, int C, int Ho, int Wo, const float* scale, const float* shift,
                                  uint16_t* dx, float* dgamma_f32, float* dbeta_f32, void* stream);
/* Global average pool NHWC bf16 -> [N,C] fp32, and backward -> bf16. */
int isic_avgpool_fwd_bf16(const uint16_t* x, float* y, int N, int HW, int C, void* stream);
int isic_avgpool_bwd_bf16(const float* dy, uint16_t* dx, int N, int HW, int C, void* stream);
/* a += b on bf16 tensors (gradient join of a residual block); n % 8 == 0. */
int isic_add_bf16(uint16_t* a, const uint16_t* b, int64_t n, void* stream);

/* ---------------------------------------------------------------- ViT-S/16 patch encoder, fp16 (BASELINE.json configs[4])
 * The reference's frozen encoder is an un-vendored ConvMAE conv-ViT (save_latent.py:42-60: eval(), no_grad,
 * forward(images, mask_ratio=0) -> latent[B,196,768]); these are the forward pieces of the ViT-S/16 this build puts in
 * its place (timm layout: pre-norm blocks, no class token here).  fp16 tensors travel as uint16_t bit patterns.
 *
 * C[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ residual): nn.Linear with its epilogue fused.  act: 0 none, 1 erf-GELU
 * (no residual with it).  residual_rows == 0: residual[M,N]; > 0: residual[residual_rows,N] broadcast over
 * m % residual_rows (the position embedding of the patch projection).  K % 64 == 0, N % 128 == 0, else UNSUPPORTED. */
int isic_gemm_f16(const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual, uint16_t* C,
                  int M, int N, int K, int act, int residual_rows, void* stream);
/* The pre-norm LayerNorms of a transformer block without their own pass over the activations (round 3).
 * isic_gemm_f16_stats: isic_gemm_f16 with a residual (act 0) that also writes, per row, the partial sums of its ROUNDED
 * outputs: row_stats[M][2 * N / 128][2] = (sum, sum of squares) over each 64-column group, i.e. everything a LayerNorm over
 * the N columns of C needs.  row_stats == NULL: plain isic_gemm_f16.
 * isic_gemm_f16_ln: C = act(LayerNorm(X) . W^T + bias) computed from the RAW rows X[M,K] (LayerNorm over K, eps):
 *   C[m][n] = act(rstd_m (sum_k X[m][k] Wg[n][k] - mean_m ln_c[n]) + bias_b[n])
 * with Wg = W . diag(gamma) (fp16), ln_c[n] = sum_k Wg[n][k], bias_b = bias + W . beta (prepared once per weight set by
 * the caller; isic_hip/vit.py).  ln_stats: ln_parts == 0 -> [M][2] = (mean, rstd) as isic_row_stats_f16 writes them;
 * ln_parts > 0 -> [M][ln_parts][2] partial sums as isic_gemm_f16_stats writes them (added in index order; variance =
 * E[x^2] - mean^2 in fp32, clamped at 0).  K % 64 == 0, K >= 128, N % 128 == 0, else UNSUPPORTED. */
int isic_gemm_f16_stats(const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual, uint16_t* C,
                        float* row_stats, int M, int N, int K, int act, int residual_rows, void* stream);
int isic_gemm_f16_ln(const uint16_t* X, const uint16_t* Wg, const float* bias_b, const float* ln_c, const float* ln_stats,
                     int ln_parts, uint16_t* C, int M, int N, int K, int act, float eps, void* stream);
/* (mean, rstd) of every row of x[M][N] (fp16, N in {128, 256, 384, 512}) with isic_layernorm_f16's arithmetic. */
int isic_row_stats_f16(const uint16_t* x, float* stats, int64_t M, int N, float eps, void* stream);
/* images NCHW fp32 -> rows[N*(H/P)*(W/P)][C*P*P] fp16: the im2col of the P x P / stride P patch projection
 * (Conv2d weight [D][C][P][P] flattened is the Linear weight).  P % 8 == 0, H % P == W % P == 0. */
int isic_vit_patchify_f16(const float* images_nchw, uint16_t* rows, int N, int C, int H, int W, int P, void* stream);
/* LayerNorm over rows of N in {128, 256, 384, 512} fp16 values, fp32 arithmetic; writes y (fp16) and / or y_f32. */
int isic_layernorm_f16(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y, float* y_f32, int64_t M,
                       int N, float eps, void* stream);
/* Multi-head self-attention over the tokens of one image: qkv[n_images*tokens][3*heads*64] (q | k | v, head-major
 * inside each) -> out[n_images*tokens][heads*64] = softmax(q k^T / 8) v per (image, head).  head_dim 64, tokens <= 208. */
int isic_attention_f16(const uint16_t* qkv, uint16_t* out, int n_images, int tokens, int heads, int head_dim,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISIC_HIP_H */
