// GraphMIL's classifier head + loss, forward AND backward, in two launches (05_train_gnns.py:136-139 classifier_light =
// Sequential(Linear(H, D), ReLU, Dropout, Linear(D, C)); :213-217 softmax; :344 F.cross_entropy(log(p + 1e-9), y)):
//
//   h1 = dropout(relu(z W1^T + b1));  logits = h1 W2^T + b2;  p = softmax(logits);  loss = mean_b CE(log(p + 1e-9), y)
//
// Per optimizer step the head works on [graphs x 128] matrices -- 25 MFLOP -- but as separate operators it is 16 dependent
// launches (two Linears, ReLU/dropout, softmax, the loss, their backwards, two weight-gradient GEMMs with split-K
// reductions, two bias column sums), each at the ~5 us floor of a dependent kernel: 107 us of a 1.3 ms step.  Here
//   isic_graph_head_fwd_bwd: a block per RH rows computes the forward of its rows, the loss per row, the gradient of every
//     activation (for d loss = 1) down to dz, and its rows' contribution to dW1 / db1 / dW2 / db2 -> workspace; the last
//     block to finish adds the per-row losses in row order (the mean does not depend on which block that is);
//   isic_graph_head_param_grads: the blocks' contributions, added in block order, times d loss, (+)= the gradients.
// Plain fp32 FMAs out of LDS: at 8 rows per block the products are too small for matrix tiles to matter.
#include "common.h"

namespace {

constexpr int RH = 8;                    // rows per block (4: the kernel 2 us shorter, the contributions pass 8 us longer)
constexpr int HT = 1024;                 // threads per block (four waves per SIMD: the scalar LDS reads need the occupancy)
constexpr int MAXC = 16;

struct HeadArgs {
  const float* z; const float* W1; const float* b1; const float* W2; const float* b2; const int64_t* labels;
  float* probs; float* loss_ps; float* loss_mean; float* dz; float* partial; unsigned int* counter;
  int B, H, D, C;
  unsigned int thr; float scale; unsigned long long seed, stream_id; const unsigned long long* clock;
};

// LDS (floats): W1s[D][H] | W1t[H][D + 1] | zc[RH][H] | h1[RH][D] | g1[RH][D] | W2s[C][D] | lg[RH][MAXC] | dl[RH][MAXC] | red[HT]
__global__ __launch_bounds__(HT) void graph_head_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int H = a.H, D = a.D, C = a.C, tid = threadIdx.x;
  float* W1s = sm;
  float* W1t = W1s + D * H;
  float* zc = W1t + H * (D + 1);
  float* h1 = zc + RH * H;
  float* g1 = h1 + RH * D;
  float* W2s = g1 + RH * D;
  float* lg = W2s + C * D;
  float* dl = lg + RH * MAXC;
  float* red = dl + RH * MAXC;
  const int r0 = blockIdx.x * RH;
  const int nr = min(RH, a.B - r0);
  unsigned long long stream_id = a.stream_id;
  if (a.clock) stream_id += a.clock[0] * 1024ULL;          // device step clock (captured graphs)

  // (eight loads in flight per thread: a load -> LDS store loop waits a full L2 round trip per element -- 32 of them for W1)
  for (int i0 = tid; i0 < D * H; i0 += 8 * HT) {
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = i0 + k * HT < D * H ? a.W1[i0 + k * HT] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * HT;
      if (i < D * H) {
        const int d = i / H, h = i - d * H;
        W1s[i] = w[k];
        W1t[h * (D + 1) + d] = w[k];
      }
    }
  }
  {
    float w[2], zv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * HT;
      w[k] = i < C * D ? a.W2[i] : 0.f;
      const int r = i / H;
      zv[k] = (i < RH * H && r < nr) ? a.z[(size_t)(r0 + r) * H + (i - r * H)] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * HT;
      if (i < C * D) W2s[i] = w[k];
      if (i < RH * H) zc[i] = zv[k];
    }
    for (int i = tid + 2 * HT; i < C * D; i += HT) W2s[i] = a.W2[i];
    for (int i = tid + 2 * HT; i < RH * H; i += HT) {
      const int r = i / H;
      zc[i] = r < nr ? a.z[(size_t)(r0 + r) * H + (i - r * H)] : 0.f;
    }
  }
  __syncthreads();

  // ---- h1 = dropout(relu(z W1^T + b1)): element (r, d), dropout word of element (r0 + r) * D + d
  for (int o = tid; o < RH * D; o += HT) {
    const int r = o / D, d = o - r * D;
    float s = a.b1[d];
    const float* zr = zc + r * H;
#pragma unroll 16
    for (int h = 0; h < H; ++h) s += zr[h] * W1t[h * (D + 1) + d];
    float v = fmaxf(s, 0.f);
    if (a.thr) v = philox_word((unsigned long long)(r0 + r) * D + d, a.seed, stream_id) >= a.thr ? v * a.scale : 0.f;
    h1[o] = v;
  }
  __syncthreads();
  // ---- logits
  for (int o = tid; o < RH * C; o += HT) {
    const int r = o / C, c = o - r * C;
    float s = a.b2[c];
#pragma unroll 16
    for (int d = 0; d < D; ++d) s += h1[r * D + d] * W2s[c * D + d];
    lg[r * MAXC + c] = s;
  }
  __syncthreads();
  // ---- softmax, loss, d logits (one thread per row; the arithmetic of softmax_rows_fwd / cross_entropy (mode 1) /
  //      softmax_rows_bwd of rowwise.hip, d loss = 1)
  if (tid < nr) {
    const int r = tid;
    const long long yl = a.labels[r0 + r];
    const bool bad_label = yl < 0 || yl >= C;             // a label outside [0, C): the row's loss (and the mean) is NaN
    const int y = bad_label ? 0 : (int)yl;
    float p[MAXC], dp[MAXC];
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lg[r * MAXC + c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(lg[r * MAXC + c] - mx);
    for (int c = 0; c < C; ++c) { p[c] = expf(lg[r * MAXC + c] - mx) / se; a.probs[(size_t)(r0 + r) * C + c] = p[c]; }
    float qm = -INFINITY;
    for (int c = 0; c < C; ++c) qm = fmaxf(qm, logf(p[c] + 1e-9f));
    float qs = 0.f;
    for (int c = 0; c < C; ++c) qs += expf(logf(p[c] + 1e-9f) - qm);
    const float lse = qm + logf(qs);
    const float lossr = bad_label ? NAN : lse - logf(p[y] + 1e-9f);
    a.loss_ps[r0 + r] = lossr;
    lg[r * MAXC + MAXC - 1] = lossr;                       // (C < MAXC: the slot is free) for the block's loss sum
    const float gs = 1.f / (float)a.B;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) {
      dp[c] = (expf(logf(p[c] + 1e-9f) - lse) - (c == y ? 1.f : 0.f)) * gs / (p[c] + 1e-9f);
      dot += p[c] * dp[c];
    }
    for (int c = 0; c < C; ++c) dl[r * MAXC + c] = p[c] * (dp[c] - dot);
  } else if (tid < RH) {
    for (int c = 0; c < C; ++c) dl[tid * MAXC + c] = 0.f;
  }
  __syncthreads();
  // ---- g1 = d(pre-activation of the first Linear) = (h1 > 0 ? scale : 0) * (d logits W2)
  for (int o = tid; o < RH * D; o += HT) {
    const int r = o / D, d = o - r * D;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dl[r * MAXC + c] * W2s[c * D + d];
    g1[o] = h1[o] > 0.f ? s * (a.thr ? a.scale : 1.f) : 0.f;
  }
  __syncthreads();
  // ---- dz = g1 W1
  for (int o = tid; o < RH * H; o += HT) {
    const int r = o / H, h = o - r * H;
    float s = 0.f;
#pragma unroll 16
    for (int d = 0; d < D; ++d) s += g1[r * D + d] * W1s[d * H + h];
    if (r < nr) a.dz[(size_t)(r0 + r) * H + h] = s;
  }
  // ---- this block's rows in the parameter gradients: [dW1 D*H | db1 D | dW2 C*D | db2 C]
  float* part = a.partial + (size_t)blockIdx.x * (D * H + D + C * D + C);
  for (int o = tid; o < D * H; o += HT) {
    const int d = o / H, h = o - d * H;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RH; ++r) s += g1[r * D + d] * zc[r * H + h];
    part[o] = s;
  }
  for (int d = tid; d < D; d += HT) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RH; ++r) s += g1[r * D + d];
    part[D * H + d] = s;
  }
  for (int o = tid; o < C * D; o += HT) {
    const int c = o / D, d = o - c * D;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RH; ++r) s += dl[r * MAXC + c] * h1[r * D + d];
    part[D * H + D + o] = s;
  }
  for (int c = tid; c < C; c += HT) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RH; ++r) s += dl[r * MAXC + c];
    part[D * H + D + C * D + c] = s;
  }
  // ---- the mean loss: every block parks the sum of its rows' losses (row order), the block that finishes last adds the
  //      blocks' sums in block order.  Device-scope ATOMICS only (performed at the coherent level): a __threadfence() pair
  //      per block writes back / invalidates the XCD's whole L2 -- 30 us of this kernel when it was written that way.
  if (tid == 0) {
    float bs = 0.f;
    for (int r = 0; r < nr; ++r) bs += lg[r * MAXC + MAXC - 1];                       // (the row's loss, parked next to its logits)
    float* slot = a.partial + (size_t)gridDim.x * (D * H + D + C * D + C) + blockIdx.x;
    (void)__hip_atomic_exchange(slot, bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the exchange has returned = it is performed, before the ticket
    unsigned int t = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      float acc = 0.f;
      const float* base = a.partial + (size_t)gridDim.x * (D * H + D + C * D + C);
      for (unsigned int b = 0; b < gridDim.x; ++b)
        acc += __hip_atomic_load(base + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.loss_mean[0] = acc / (float)a.B;
      __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// out[i] = (accumulate ? out[i] : 0) + gscale * sum over the blocks (in block order) of partial[b][off + i]: four outputs
__global__ __launch_bounds__(256) void graph_head_grads_kernel(const float* __restrict__ partial, int nblocks, int stride,
                                                                const float* __restrict__ gscale, float* dW1, float* db1,
                                                                float* dW2, float* db2, int n1, int n2, int n3, int n4,
                                                                int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= stride) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * stride + i];
  if (gscale) s *= gscale[0];
  float* o = i < n1 ? dW1 + i : i < n1 + n2 ? db1 + (i - n1) : i < n1 + n2 + n3 ? dW2 + (i - n1 - n2) : db2 + (i - n1 - n2 - n3);
  *o = accumulate ? *o + s : s;
}

size_t head_lds_bytes(int H, int D, int C) {
  return sizeof(float) * ((size_t)D * H + (size_t)H * (D + 1) + (size_t)RH * H + 2 * (size_t)RH * D + (size_t)C * D +
                          2 * RH * MAXC + HT);
}

}  // namespace

extern "C" {

int isic_graph_head_supported(int H, int D, int C) {
  ISIC_CHECK_ARG(H > 0 && D > 0 && C > 0);
  if (C >= MAXC) return ISIC_ERR_UNSUPPORTED;
  return head_lds_bytes(H, D, C) <= 160 * 1024 - 256 ? ISIC_OK : ISIC_ERR_UNSUPPORTED;   // (the kernel has a few static bytes of its own)
}

size_t isic_graph_head_workspace_bytes(int B, int H, int D, int C) {
  if (B <= 0 || H <= 0 || D <= 0 || C <= 0) return 0;
  return (size_t)ceil_div(B, RH) * ((size_t)D * H + D + (size_t)C * D + C + 1) * sizeof(float);   // + the block's loss sum
}

int isic_graph_head_fwd_bwd(const float* z, const float* W1, const float* b1, const float* W2, const float* b2,
                            const int64_t* labels, int B, int H, int D, int C, uint32_t drop_threshold, float drop_scale,
                            uint64_t seed, uint64_t stream_id, const uint64_t* clock, float* probs, float* loss_per_sample,
                            float* loss_mean, float* dz, void* workspace, size_t workspace_bytes, uint32_t* counter,
                            void* stream) {
  ISIC_CHECK_ARG(B > 0 && H > 0 && D > 0 && C > 0);
  ISIC_CHECK_ARG(z && W1 && b1 && W2 && b2 && labels && probs && loss_per_sample && loss_mean && dz && workspace && counter);
  if (isic_graph_head_supported(H, D, C) != ISIC_OK) return ISIC_ERR_UNSUPPORTED;
  const size_t lds = head_lds_bytes(H, D, C);
  if (workspace_bytes < isic_graph_head_workspace_bytes(B, H, D, C) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return ISIC_ERR_WORKSPACE;
  static IsicPerDeviceOnce once;
  if (isic_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(graph_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024 - 256);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  HeadArgs a;
  a.z = z; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.labels = labels;
  a.probs = probs; a.loss_ps = loss_per_sample; a.loss_mean = loss_mean; a.dz = dz;
  const int nb = ceil_div(B, RH);
  a.counter = counter;                                     // zero before the first launch; the kernel leaves it zero
  a.partial = reinterpret_cast<float*>(workspace);         // [nb][D*H + D + C*D + C] contributions
  a.B = B; a.H = H; a.D = D; a.C = C;
  a.thr = drop_threshold; a.scale = drop_scale; a.seed = seed; a.stream_id = stream_id;
  a.clock = reinterpret_cast<const unsigned long long*>(clock);
  hipLaunchKernelGGL(graph_head_kernel, dim3(nb), dim3(HT), lds, as_stream(stream), a);
  return isic_launch_status();
}

int isic_graph_head_param_grads(const void* workspace, int B, int H, int D, int C, const float* grad_scale, float* dW1,
                                float* db1, float* dW2, float* db2, int accumulate, void* stream) {
  ISIC_CHECK_ARG(B > 0 && H > 0 && D > 0 && C > 0 && workspace && dW1 && db1 && dW2 && db2);
  const int stride = D * H + D + C * D + C;
  hipLaunchKernelGGL(graph_head_grads_kernel, dim3(ceil_div(stride, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float*>(workspace), ceil_div(B, RH),
                     stride, grad_scale, dW1, db1, dW2, db2, D * H, D, C * D, C, accumulate);
  return isic_launch_status();
}

}  // extern "C"
