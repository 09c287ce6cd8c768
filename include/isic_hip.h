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

/* out[n] = sum_m X[m,n] (+ beta*out): bias gradients of the layers above. */
int isic_colsum_f32(const float* X, int M, int N, int ldx, float* out, float beta, void* stream);

/* y = act'(...) helpers for backward: dx = dy * (1 - t*t)  (tanh, from its output t). */
int isic_tanh_bwd_f32(const float* dy, const float* t, float* dx, int64_t n, void* stream);

/* y = dropout(relu(x)) in place on x (x already holds the pre-activation);
 * replaces nn.ReLU + nn.Dropout of utils_g_mil.py:50-52.  bwd: dx = dy * mask,
 * mask recomputed from y > 0 (kept & positive) and the same counter stream. */
int isic_relu_dropout_fwd_f32(float* x, int64_t n, uint32_t drop_threshold, float drop_scale, uint64_t seed,
                              uint64_t stream_id, void* stream);
int isic_relu_dropout_bwd_f32(const float* y, float* dy, int64_t n, float drop_scale, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* ISIC_HIP_H */
