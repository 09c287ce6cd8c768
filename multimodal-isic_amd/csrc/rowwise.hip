// Row-wise fused ops (gfx950): LayerNorm(+ReLU+dropout+residual) fwd/bwd, cross
// entropy, row softmax, flat Adam/AdamW.  All HBM-bound; one wave per row with
// 64-lane shuffle reductions.
//
// Replaces 05_train_gnns.py:187-199 (LayerNorm -> ReLU -> Dropout -> residual),
// model.py:75-82 (Linear -> LayerNorm -> ReLU -> Dropout), the losses at
// 01_train_mil_teacher.py:143,244 / 05_train_gnns.py:344 and
// torch.optim.AdamW/Adam (01:217-224, 05:332-333).
#include "common.h"

namespace {

__device__ __forceinline__ bool keep_elem(unsigned long long idx, unsigned int thr, unsigned long long seed,
                                          unsigned long long stream_id) {
  return thr == 0 || philox_word(idx, seed, stream_id) >= thr;
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ residual, float* __restrict__ y, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, int M, int N, float eps, int relu, unsigned int thr, float scale,
    unsigned long long seed, unsigned long long stream_id,
    const unsigned long long* __restrict__ clock) {
  if (clock) stream_id += clock[0] * 1024ULL;        // device step clock (captured graphs): stream = base + step * 1024
  const int lane = threadIdx.x & 63;
  const int row0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int row = row0; row < M; row += gridDim.x * 4) {
    const float* xr = x + (size_t)row * N;
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += xr[j];
    const float mean = wave_sum(s) / (float)N;
    float v = 0.f;
    for (int j = lane; j < N; j += 64) { const float d = xr[j] - mean; v += d * d; }
    const float rstd = rsqrtf(wave_sum(v) / (float)N + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    for (int j = lane; j < N; j += 64) {
      float o = (xr[j] - mean) * rstd * gamma[j] + beta[j];
      if (relu) o = fmaxf(o, 0.f);
      if (thr) o = keep_elem((unsigned long long)row * N + j, thr, seed, stream_id) ? o * scale : 0.f;
      if (residual) o += residual[(size_t)row * N + j];
      y[(size_t)row * N + j] = o;
    }
  }
}

// dgamma/dbeta: per-lane register partials over the rows of this block (JN columns per lane), then
// one LDS tree over the 4 waves and a single fp32 atomic per column per block.
template <int JN>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int N, int relu,
    unsigned int thr, float scale, unsigned long long seed, unsigned long long stream_id,
    const unsigned long long* __restrict__ clock, float* __restrict__ partial) {
  if (clock) stream_id += clock[0] * 1024ULL;        // device step clock (captured graphs): stream = base + step * 1024
  __shared__ float red[2][4][64 * JN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pg[JN], pb[JN];
#pragma unroll
  for (int j = 0; j < JN; ++j) { pg[j] = 0.f; pb[j] = 0.f; }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    float g[JN], xh[JN];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int col = lane + 64 * j;
      g[j] = 0.f; xh[j] = 0.f;
      if (col < N) {
        const size_t idx = (size_t)row * N + col;
        xh[j] = (x[idx] - mean) * rstd;
        float gg = dy[idx];
        if (thr) gg = keep_elem((unsigned long long)idx, thr, seed, stream_id) ? gg * scale : 0.f;
        if (relu && !(xh[j] * gamma[col] + beta[col] > 0.f)) gg = 0.f;
        pg[j] += gg * xh[j];
        pb[j] += gg;
        g[j] = gg * gamma[col];
        s1 += g[j];
        s2 += g[j] * xh[j];
      }
    }
    s1 = wave_sum(s1) / (float)N;
    s2 = wave_sum(s2) / (float)N;
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int col = lane + 64 * j;
      if (col < N) dx[(size_t)row * N + col] = rstd * (g[j] - s1 - xh[j] * s2);
    }
  }
#pragma unroll
  for (int j = 0; j < JN; ++j) { red[0][wave][lane + 64 * j] = pg[j]; red[1][wave][lane + 64 * j] = pb[j]; }
  __syncthreads();
  for (int col = threadIdx.x; col < N; col += 256) {
    const float tg = (red[0][0][col] + red[0][1][col]) + (red[0][2][col] + red[0][3][col]);
    const float tb = (red[1][0][col] + red[1][1][col]) + (red[1][2][col] + red[1][3][col]);
    if (partial) {                                   // deterministic: the block's own row, ln_bwd_reduce_kernel adds in order
      partial[(size_t)blockIdx.x * 2 * N + col] = tg;
      partial[(size_t)blockIdx.x * 2 * N + N + col] = tb;
    } else {
      atomicAdd(&dgamma[col], tg);
      atomicAdd(&dbeta[col], tb);
    }
  }
}

// dgamma[n] += sum over the blocks' partial rows (and dbeta likewise), in block order: 16 lanes per column, fixed xor tree
// (dxsum != NULL: a third row per block, the column sums of dx -- the bias gradient of the layer below)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partial, int nblocks, int N,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ dxsum) {
  const int g = threadIdx.x & 15;
  const int nv = dxsum ? 3 : 2;
  const int n = blockIdx.x * 16 + (threadIdx.x >> 4);        // 0 .. nv*N-1: dgamma columns, then dbeta columns (, then dx sums)
  float s = 0.f;
  if (n < nv * N)
    for (int b = g; b < nblocks; b += 16) s += partial[(size_t)b * nv * N + n];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
  if (n < nv * N && g == 0) {
    if (n < N) dgamma[n] += s; else if (n < 2 * N) dbeta[n - N] += s; else dxsum[n - 2 * N] += s;
  }
}

// ---- vector forms for N = 4 * LPR, LPR in {16, 32, 64} (the GNN hidden sizes 64 / 128 / 256): a lane owns 4 consecutive
// columns (16-byte accesses), LPR lanes a row, so a wave works on 64 / LPR rows at once (many short waves instead of
// a few long ones); one Philox block per 4 elements (the scalar kernels recompute it per element).  The wave-per-row
// kernels above are latency-bound on their load -> reduce -> store chain (55 us for 50 k x 128 backward; 13 us of traffic).
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void keep4(bool (&k)[4], unsigned long long idx, unsigned int thr, unsigned long long seed,
                                      unsigned long long stream_id) {
  const Philox4 r = philox_block(idx >> 2, seed, stream_id);     // idx is a multiple of 4
  k[0] = r.x >= thr; k[1] = r.y >= thr; k[2] = r.z >= thr; k[3] = r.w >= thr;
}

template <int LPR>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ residual, float* __restrict__ y, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, int M, float eps, int relu, unsigned int thr, float scale,
    unsigned long long seed, unsigned long long stream_id,
    const unsigned long long* __restrict__ clock) {
  if (clock) stream_id += clock[0] * 1024ULL;        // device step clock (captured graphs): stream = base + step * 1024
  constexpr int RPW = 64 / LPR, N = 4 * LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, col = (lane % LPR) * 4;
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + col), b4 = *reinterpret_cast<const f32x4*>(beta + col);
  const int stride = gridDim.x * 4 * RPW;
  for (int base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW; base < M; base += stride) {
    const bool valid = base + sub < M;
    const int row = valid ? base + sub : M - 1;
    const size_t idx = (size_t)row * N + col;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + idx);
    f32x4 rv = {0.f, 0.f, 0.f, 0.f};
    if (residual) rv = *reinterpret_cast<const f32x4*>(residual + idx);
    const float mean = group_sum<LPR>((xv[0] + xv[1]) + (xv[2] + xv[3])) * (1.f / N);
    const f32x4 d = xv - mean;
    const float rstd = rsqrtf(group_sum<LPR>((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.f / N) + eps);
    f32x4 o = d * rstd * g4 + b4;
    if (relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
    if (thr) {
      bool k[4];
      keep4(k, (unsigned long long)idx, thr, seed, stream_id);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = k[e] ? o[e] * scale : 0.f;
    }
    o += rv;
    if (valid) {
      *reinterpret_cast<f32x4*>(y + idx) = o;
      if (col == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
  }
}

// 1024-thread blocks, one per CU at most: the dgamma / dbeta atomics of all blocks hit the same 2 N addresses and retire
// at ~3 ns each per cache line (1568 blocks of 256 threads: 160 us, all atomics), so there are few, fat blocks.
template <int LPR, bool DXS>
__global__ __launch_bounds__(1024) void layernorm_bwd_vec_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int relu,
    unsigned int thr, float scale, unsigned long long seed, unsigned long long stream_id,
    const unsigned long long* __restrict__ clock, float* __restrict__ partial) {
  if (clock) stream_id += clock[0] * 1024ULL;        // device step clock (captured graphs): stream = base + step * 1024
  constexpr int RPW = 64 / LPR, N = 4 * LPR, NV = DXS ? 3 : 2;
  __shared__ f32x4 red[NV][1024];
  const int lane = threadIdx.x & 63, sub = lane / LPR, col = (lane % LPR) * 4;
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + col), b4 = *reinterpret_cast<const f32x4*>(beta + col);
  f32x4 pg = {0.f, 0.f, 0.f, 0.f}, pb = pg, pd = pg;
  const int stride = gridDim.x * 16 * RPW;
  for (int base = (blockIdx.x * 16 + (threadIdx.x >> 6)) * RPW; base < M; base += stride) {
    const bool valid = base + sub < M;
    const int row = valid ? base + sub : M - 1;
    const size_t idx = (size_t)row * N + col;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + idx);
    f32x4 gg = *reinterpret_cast<const f32x4*>(dy + idx);
    const float mean = mean_in[row], rstd = rstd_in[row];
    const f32x4 xh = (xv - mean) * rstd;
    if (thr) {
      bool k[4];
      keep4(k, (unsigned long long)idx, thr, seed, stream_id);
#pragma unroll
      for (int e = 0; e < 4; ++e) gg[e] = k[e] ? gg[e] * scale : 0.f;
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(xh[e] * g4[e] + b4[e] > 0.f)) gg[e] = 0.f;
    }
    if (!valid) gg = (f32x4){0.f, 0.f, 0.f, 0.f};
    pg += gg * xh;
    pb += gg;
    const f32x4 g = gg * g4;
    const float s1 = group_sum<LPR>((g[0] + g[1]) + (g[2] + g[3])) * (1.f / N);
    const f32x4 gx = g * xh;
    const float s2 = group_sum<LPR>((gx[0] + gx[1]) + (gx[2] + gx[3])) * (1.f / N);
    const f32x4 dxv = (g - s1 - xh * s2) * rstd;
    if (valid) *reinterpret_cast<f32x4*>(dx + idx) = dxv;
    if (DXS && valid) pd += dxv;
  }
  red[0][threadIdx.x] = pg;
  red[1][threadIdx.x] = pb;
  if (DXS) red[NV - 1][threadIdx.x] = pd;
  __syncthreads();
  if (threadIdx.x < LPR) {                     // the 16 waves x RPW row slots that hold the same columns
    f32x4 tg = {0.f, 0.f, 0.f, 0.f}, tb = tg, td = tg;
#pragma unroll
    for (int k = 0; k < 16 * RPW; ++k) {
      tg += red[0][k * LPR + threadIdx.x]; tb += red[1][k * LPR + threadIdx.x];
      if (DXS) td += red[NV - 1][k * LPR + threadIdx.x];
    }
    if (partial) {                                   // deterministic: the block's own row of [NV][N]
      *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.x * NV * N + threadIdx.x * 4) = tg;
      *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.x * NV * N + N + threadIdx.x * 4) = tb;
      if (DXS) *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.x * NV * N + 2 * N + threadIdx.x * 4) = td;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(&dgamma[threadIdx.x * 4 + e], tg[e]);
        atomicAdd(&dbeta[threadIdx.x * 4 + e], tb[e]);
      }
    }
  }
}

// y = x / max(||x||_2, eps) per row (F.normalize(p=2, dim=-1) of SAGEConv(normalize=True)), wave per row
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          float* __restrict__ norm, int M, int N, float eps) {
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    float s = 0.f;
    for (int j = lane; j < N; j += 64) { const float v = x[(size_t)row * N + j]; s += v * v; }
    const float n = fmaxf(sqrtf(wave_sum(s)), eps);
    if (lane == 0) norm[row] = n;
    for (int j = lane; j < N; j += 64) y[(size_t)row * N + j] = x[(size_t)row * N + j] / n;
  }
}
// dx = (dy - y * (y . dy)) / n   (rows whose norm was clamped get dx = dy / eps, like torch)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          const float* __restrict__ norm, float* __restrict__ dx,
                                                          int M, int N, float eps) {
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    const float n = norm[row];
    float d = 0.f;
    if (n > eps)
      for (int j = lane; j < N; j += 64) d += y[(size_t)row * N + j] * dy[(size_t)row * N + j];
    d = wave_sum(d);
    for (int j = lane; j < N; j += 64)
      dx[(size_t)row * N + j] = (dy[(size_t)row * N + j] - y[(size_t)row * N + j] * d) / n;
  }
}

// one block; thread per sample; deterministic block tree for the mean
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ in,
                                                             const int64_t* __restrict__ labels, int B, int C, int mode,
                                                             float grad_scale, float* __restrict__ loss_ps,
                                                             float* __restrict__ loss_mean, float* __restrict__ d_in) {
  __shared__ float part[256];
  float acc = 0.f;
  const float gs = grad_scale / (float)B;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* r = in + (size_t)b * C;
    const int y = (int)labels[b];
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) { const float q = mode ? logf(r[c] + 1e-9f) : r[c]; mx = fmaxf(mx, q); }
    float se = 0.f;
    for (int c = 0; c < C; ++c) { const float q = mode ? logf(r[c] + 1e-9f) : r[c]; se += expf(q - mx); }
    const float lse = mx + logf(se);
    const float qy = mode ? logf(r[y] + 1e-9f) : r[y];
    const float loss = lse - qy;
    if (loss_ps) loss_ps[b] = loss;
    acc += loss;
    if (d_in) {
      for (int c = 0; c < C; ++c) {
        const float q = mode ? logf(r[c] + 1e-9f) : r[c];
        float d = (expf(q - lse) - (c == y ? 1.f : 0.f)) * gs;
        if (mode) d /= (r[c] + 1e-9f);
        d_in[(size_t)b * C + c] = d;
      }
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss_mean) loss_mean[0] = part[0] / (float)B;
}

__global__ void softmax_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ p, int M, int N) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= M) return;
  const float* r = x + (size_t)row * N;
  float mx = -INFINITY;
  for (int c = 0; c < N; ++c) mx = fmaxf(mx, r[c]);
  float se = 0.f;
  for (int c = 0; c < N; ++c) se += expf(r[c] - mx);
  for (int c = 0; c < N; ++c) p[(size_t)row * N + c] = expf(r[c] - mx) / se;
}

__global__ void softmax_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                        float* __restrict__ dx, int M, int N) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= M) return;
  float dot = 0.f;
  for (int c = 0; c < N; ++c) dot += p[(size_t)row * N + c] * dp[(size_t)row * N + c];
  for (int c = 0; c < N; ++c) dx[(size_t)row * N + c] = p[(size_t)row * N + c] * (dp[(size_t)row * N + c] - dot);
}

// torch.optim.AdamW single-tensor order of operations (fp32), see header.
// clock[0]: dropout step (stream id = base + clock[0] * 1024), clock[1]: optimizer steps taken.  One thread: the LAST
// kernel of a captured step, so that every kernel of the step read the same values.
__global__ void step_clock_advance_kernel(unsigned long long* __restrict__ clock, int dsteps, int osteps) {
  clock[0] += (unsigned long long)dsteps;
  clock[1] += (unsigned long long)osteps;
}

__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, int64_t n, float step_size, float beta1, float beta2,
                                 float eps, float decay_factor, float l2, float bc2_sqrt, float grad_scale,
                                 unsigned short* __restrict__ p_bf16, const unsigned long long* __restrict__ clock,
                                 double lr) {
  if (clock) {
    // device step clock (captured graphs): t = clock[1] + 1; step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t),
    // formed in fp64 as the host does (optim.py)
    const double t = (double)(clock[1] + 1ULL);
    step_size = (float)(lr / (1.0 - pow((double)beta1, t)));     // one rounding, as the host's lr / bc1
    bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, t));
  }
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float w1 = 1.f - beta1, w2 = 1.f - beta2;
  for (; i < n; i += stride) {
    float pv = p[i], gv = g[i] * grad_scale, mv = m[i], vv = v[i];
    pv = pv * decay_factor;                         // AdamW: param.mul_(1 - lr * weight_decay)
    if (l2 != 0.f) gv = gv + l2 * pv;               // Adam:  grad.add(param, alpha=weight_decay)
    mv = mv + w1 * (gv - mv);                       // exp_avg.lerp_(grad, 1 - beta1)
    vv = vv * beta2 + (w2 * gv) * gv;               // mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pv = pv - step_size * (mv / denom);             // addcdiv_(exp_avg, denom, value=-step_size)
    p[i] = pv; m[i] = mv; v[i] = vv;
    if (p_bf16) p_bf16[i] = f32_to_bf16_bits(pv);
  }
}

}  // namespace

extern "C" {

int isic_layernorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                       float* mean, float* rstd, int M, int N, float eps, int relu, uint32_t drop_threshold,
                       float drop_scale, uint64_t seed, uint64_t stream_id, void* stream) {
  return isic_layernorm_fwd_clk(x, gamma, beta, residual, y, mean, rstd, M, N, eps, relu, drop_threshold, drop_scale, seed,
                                stream_id, nullptr, stream);
}

int isic_layernorm_fwd_clk(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                           float* mean, float* rstd, int M, int N, float eps, int relu, uint32_t drop_threshold,
                           float drop_scale, uint64_t seed, uint64_t stream_id, const uint64_t* clock, void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N > 0);
  if (M == 0) return ISIC_OK;
  ISIC_CHECK_ARG(x && gamma && beta && y && mean && rstd);
  const bool al16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) |
                      reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
  if (al16 && (N == 64 || N == 128 || N == 256)) {
    const int rpw = 256 / N;                                       // rows per wave pass
    int gridv = ceil_div(M, 4 * rpw * 2);
    if (gridv > 4096) gridv = 4096;
#define LAUNCH_LNF(LPR)                                                                                              \
  hipLaunchKernelGGL(layernorm_fwd_vec_kernel<LPR>, dim3(gridv), dim3(256), 0, as_stream(stream), x, gamma, beta, residual, \
                     y, mean, rstd, M, eps, relu, drop_threshold, drop_scale, (unsigned long long)seed,              \
                     (unsigned long long)stream_id, (const unsigned long long*)clock)
    if (N == 64) LAUNCH_LNF(16); else if (N == 128) LAUNCH_LNF(32); else LAUNCH_LNF(64);
#undef LAUNCH_LNF
    return isic_launch_status();
  }
  int grid = ceil_div(M, 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, gamma, beta, residual, y,
                     mean, rstd, M, N, eps, relu, drop_threshold, drop_scale, (unsigned long long)seed,
                     (unsigned long long)stream_id, (const unsigned long long*)clock);
  return isic_launch_status();
}

int isic_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, float* dx, float* dgamma, float* dbeta, int M, int N, int relu,
                       uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id, void* stream) {
  return isic_layernorm_bwd_clk(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, M, N, relu, drop_threshold, drop_scale, seed,
                                stream_id, nullptr, stream);
}

int isic_layernorm_bwd_clk(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                           const float* rstd, float* dx, float* dgamma, float* dbeta, int M, int N, int relu,
                           uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id,
                           const uint64_t* clock, void* stream) {
  return isic_layernorm_bwd_ws(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, M, N, relu, drop_threshold, drop_scale, seed,
                               stream_id, clock, nullptr, 0, stream);
}

size_t isic_layernorm_bwd_workspace_bytes(int N) {
  return N > 0 && N <= 1024 ? (size_t)1024 * 3 * N * sizeof(float) : 0;      // at most 1024 blocks, a [3][N] row each
}

int isic_layernorm_bwd_ws(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                          const float* rstd, float* dx, float* dgamma, float* dbeta, int M, int N, int relu,
                          uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id,
                          const uint64_t* clock, void* workspace, size_t workspace_bytes, void* stream) {
  return isic_layernorm_bwd_dxsum_ws(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, nullptr, M, N, relu, drop_threshold,
                                     drop_scale, seed, stream_id, clock, workspace, workspace_bytes, stream);
}

int isic_layernorm_bwd_dxsum_ws(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                                const float* rstd, float* dx, float* dgamma, float* dbeta, float* dxsum, int M, int N,
                                int relu, uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id,
                                const uint64_t* clock, void* workspace, size_t workspace_bytes, void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N > 0);
  if (M == 0) return ISIC_OK;
  ISIC_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta);
  if (N > 1024) return ISIC_ERR_UNSUPPORTED;
  // with a workspace the blocks' dgamma / dbeta partial rows are added in block order (bit-reproducible); without one
  // they meet through fp32 atomics in arrival order
  float* partial = (workspace && workspace_bytes >= isic_layernorm_bwd_workspace_bytes(N) &&
                    (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) ? reinterpret_cast<float*>(workspace) : nullptr;
  const bool al16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) |
                      reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
  if (al16 && (N == 64 || N == 128 || N == 256)) {
    const int rpw = 256 / N;
    int gridv = ceil_div(M, 16 * rpw * 4);                         // >= 4 passes per wave
    if (gridv > 192) gridv = 192;                                  // few fat blocks: see layernorm_bwd_vec_kernel
    if (dxsum && !partial) return ISIC_ERR_WORKSPACE;              // the column sums of dx exist in the order-fixed form only
#define LAUNCH_LNB(LPR, DXS)                                                                                             \
  hipLaunchKernelGGL((layernorm_bwd_vec_kernel<LPR, DXS>), dim3(gridv), dim3(1024), 0, as_stream(stream), dy, x, gamma, beta, mean, \
                     rstd, dx, dgamma, dbeta, M, relu, drop_threshold, drop_scale, (unsigned long long)seed,        \
                     (unsigned long long)stream_id, (const unsigned long long*)clock, partial)
    if (dxsum) { if (N == 64) LAUNCH_LNB(16, true); else if (N == 128) LAUNCH_LNB(32, true); else LAUNCH_LNB(64, true); }
    else { if (N == 64) LAUNCH_LNB(16, false); else if (N == 128) LAUNCH_LNB(32, false); else LAUNCH_LNB(64, false); }
#undef LAUNCH_LNB
    if (partial)
      hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(ceil_div((dxsum ? 3 : 2) * N, 16)), dim3(256), 0, as_stream(stream), partial,
                         gridv, N, dgamma, dbeta, dxsum);
    return isic_launch_status();
  }
  if (dxsum) return ISIC_ERR_UNSUPPORTED;                          // N in {64, 128, 256}, 16-byte aligned rows only
  int grid = ceil_div(M, 4 * 8);  // >= 8 rows per wave amortise the atomics
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
#define LAUNCH_LN(JN)                                                                                          \
  hipLaunchKernelGGL(layernorm_bwd_kernel<JN>, dim3(grid), dim3(256), 0, as_stream(stream), dy, x, gamma, beta, \
                     mean, rstd, dx, dgamma, dbeta, M, N, relu, drop_threshold, drop_scale,                     \
                     (unsigned long long)seed, (unsigned long long)stream_id, (const unsigned long long*)clock, partial)
  if (N <= 128) LAUNCH_LN(2);
  else if (N <= 256) LAUNCH_LN(4);
  else if (N <= 512) LAUNCH_LN(8);
  else LAUNCH_LN(16);
#undef LAUNCH_LN
  if (partial)
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(ceil_div(2 * N, 16)), dim3(256), 0, as_stream(stream), partial, grid, N,
                       dgamma, dbeta, (float*)nullptr);
  return isic_launch_status();
}

int isic_l2normalize_fwd(const float* x, float* y, float* norm, int M, int N, float eps, void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N > 0);
  if (M == 0) return ISIC_OK;
  ISIC_CHECK_ARG(x && y && norm);
  int grid = ceil_div(M, 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, y, norm, M, N, eps);
  return isic_launch_status();
}

int isic_l2normalize_bwd(const float* dy, const float* y, const float* norm, float* dx, int M, int N, float eps,
                         void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N > 0);
  if (M == 0) return ISIC_OK;
  ISIC_CHECK_ARG(dy && y && norm && dx);
  int grid = ceil_div(M, 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), dy, y, norm, dx, M, N, eps);
  return isic_launch_status();
}

int isic_cross_entropy(const float* in, const int64_t* labels, int B, int C, int mode, float grad_scale,
                       float* loss_per_sample, float* loss_mean, float* d_in, void* stream) {
  ISIC_CHECK_ARG(B > 0 && C > 0 && in && labels);
  ISIC_CHECK_ARG(mode == 0 || mode == 1);
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, as_stream(stream), in, labels, B, C, mode,
                     grad_scale, loss_per_sample, loss_mean, d_in);
  return isic_launch_status();
}

int isic_softmax_rows_fwd(const float* logits, float* probs, int M, int N, void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N > 0);
  if (M == 0) return ISIC_OK;
  ISIC_CHECK_ARG(logits && probs);
  hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3(ceil_div(M, 64)), dim3(64), 0, as_stream(stream), logits, probs, M, N);
  return isic_launch_status();
}

int isic_softmax_rows_bwd(const float* probs, const float* d_probs, float* d_logits, int M, int N, void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N > 0);
  if (M == 0) return ISIC_OK;
  ISIC_CHECK_ARG(probs && d_probs && d_logits);
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(ceil_div(M, 64)), dim3(64), 0, as_stream(stream), probs, d_probs,
                     d_logits, M, N);
  return isic_launch_status();
}

int isic_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float step_size, float beta1, float beta2,
                   float eps, float decay_factor, float l2, float bias_correction2_sqrt, float grad_scale,
                   uint16_t* p_bf16, void* stream) {
  ISIC_CHECK_ARG(n >= 0);
  if (n == 0) return ISIC_OK;
  ISIC_CHECK_ARG(p && g && m && v && bias_correction2_sqrt > 0.f);
  int64_t grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(adam_step_kernel, dim3((int)grid), dim3(256), 0, as_stream(stream), p, g, m, v, n, step_size,
                     beta1, beta2, eps, decay_factor, l2, bias_correction2_sqrt, grad_scale, p_bf16,
                     (const unsigned long long*)nullptr, 0.0);
  return isic_launch_status();
}

int isic_adam_step_clk(float* p, const float* g, float* m, float* v, int64_t n, double lr, float beta1, float beta2,
                       float eps, float decay_factor, float l2, float grad_scale, uint16_t* p_bf16, const uint64_t* clock,
                       void* stream) {
  ISIC_CHECK_ARG(n >= 0 && clock);
  if (n == 0) return ISIC_OK;
  ISIC_CHECK_ARG(p && g && m && v);
  int64_t grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(adam_step_kernel, dim3((int)grid), dim3(256), 0, as_stream(stream), p, g, m, v, n, 0.f, beta1, beta2, eps,
                     decay_factor, l2, 1.f, grad_scale, p_bf16, (const unsigned long long*)clock, lr);
  return isic_launch_status();
}

int isic_step_clock_advance(uint64_t* clock, int dropout_steps, int optimizer_steps, void* stream) {
  ISIC_CHECK_ARG(clock && dropout_steps >= 0 && optimizer_steps >= 0);
  hipLaunchKernelGGL(step_clock_advance_kernel, dim3(1), dim3(1), 0, as_stream(stream),
                     reinterpret_cast<unsigned long long*>(clock), dropout_steps, optimizer_steps);
  return isic_launch_status();
}

}  // extern "C"
