// HBM-bound companion kernels of the conv encoder (gfx950), NHWC bf16, 16-byte vector
// accesses (8 channels per lane): BatchNorm2d training-mode statistics / apply / backward,
// 3x3/2 max-pool, global average pool, layout packing, gradient joins.
//
// The reference has no counterpart (its encoder is an un-vendored ConvMAE, save_latent.py:42-60);
// these implement the ResNet-18 named by BASELINE.json configs[1] with torchvision semantics
// (BatchNorm2d eps 1e-5, momentum 0.1, biased variance for normalisation, unbiased for the running
// estimate; MaxPool2d(3, 2, 1) with first-maximum tie breaking; AdaptiveAvgPool2d(1)).
#include "common.h"
#include "pool_grad.h"

namespace {

using isic_pool::unpack8;
using isic_pool::pack8;
using isic_pool::PoolGeom;

constexpr int FLAT_APPLY = 2, FLAT_BWD = 4;      // vectors per thread of the BatchNorm apply / backward-apply passes

// ---------------------------------------------------------------- BatchNorm statistics
// thread -> channel group cg = tid % (C/8), row lane rl = tid / (C/8); per-thread fp32 partials over
// a grid-strided set of rows, LDS tree over the rows-lanes of the block, fp64 atomics to global.
template <int NACC>
__device__ __forceinline__ void block_reduce_to_global(float (&acc)[NACC][8], int C, double* const (&dst)[NACC]) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x, cgs = C >> 3, cg = tid % cgs, rl = tid / cgs, rls = 256 / cgs;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(rl * cgs + cg) * 8 + j] = acc[a][j];
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      // channel c: group c>>3, element c&7
      double s = 0.0;
      for (int r = 0; r < rls; ++r) s += (double)red[(r * cgs + (c >> 3)) * 8 + (c & 7)];
      atomicAdd(dst[a] + c, s);
    }
  }
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const unsigned short* __restrict__ x, int64_t rows, int C,
                                                        double* __restrict__ sum, double* __restrict__ sumsq) {
  const int tid = threadIdx.x, cgs = C >> 3, cg = tid % cgs, rl = tid / cgs, rls = 256 / cgs;
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  // a block sums a CONTIGUOUS range of rows (DRAM pages walked in order: 5.8 -> 6.2 TB/s for the backward sums)
  const int64_t per_block = ((rows + gridDim.x - 1) / gridDim.x + rls - 1) / rls * rls;
  const int64_t r_end = min(rows, ((int64_t)blockIdx.x + 1) * per_block);
#pragma unroll 4
  for (int64_t r = (int64_t)blockIdx.x * per_block + rl; r < r_end; r += rls) {
    float f[8];
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + r * C + cg * 8)), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[0][j] += f[j]; acc[1][j] += f[j] * f[j]; }
  }
  double* const dst[2] = {sum, sumsq};
  block_reduce_to_global<2>(acc, C, dst);
}

// 16 lanes per channel: lane sl adds the slot rows sl, sl + 16, ... (the producing convolutions leave one row per block:
// up to 256, the stem 1024), then a fixed xor tree joins the 16 partial sums -- a fixed order, so the result does not
// depend on which block finished first, and the 256-1024 dependent loads of a one-thread-per-channel loop are gone.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ sum, const double* __restrict__ sumsq,
                                                           int nslots,
                                   int64_t rows, int C, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                   float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int sl = threadIdx.x & 15;
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  const double n = (double)rows;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int k = sl; k < nslots; k += 16) { s1 += sum[(size_t)k * C + c]; s2 += sumsq[(size_t)k * C + c]; }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 16); s2 += __shfl_xor(s2, o, 16); }
  if (c >= C || sl != 0) return;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  mean_o[c] = (float)mean;
  rstd_o[c] = rstd;
  if (running_mean) {
    const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// dbeta = sum dz, dgamma = sum dz * xhat = rstd * (sum dz*y - mean * sum dz), from the per-slot partial sums a data-gradient
// convolution accumulated in its epilogue (conv_halo.hip STATS 2); fp64 throughout
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ s_dz, const double* __restrict__ s_dzy, int nslots, int C,
                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                       double* __restrict__ dgamma, double* __restrict__ dbeta) {
  const int sl = threadIdx.x & 15;                                  // 16 lanes per channel, fixed xor tree (bn_finalize_kernel)
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int s = sl; s < nslots; s += 16) { a += s_dz[(size_t)s * C + c]; b += s_dzy[(size_t)s * C + c]; }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 16); b += __shfl_xor(b, o, 16); }
  if (c >= C || sl != 0) return;
  dbeta[c] = a;
  dgamma[c] = (double)rstd[c] * (b - (double)mean[c] * a);
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

template <bool RES_AFFINE>
__global__ __launch_bounds__(256) void bn_apply_kernel(const unsigned short* __restrict__ x,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        const unsigned short* __restrict__ residual,
                                                        unsigned short* __restrict__ y,
                                                        unsigned char* __restrict__ relu_mask, int64_t nvec, int C, int relu,
                                                        const float* __restrict__ res_scale,
                                                        const float* __restrict__ res_shift) {
  // a block starts at a multiple of 256 vectors and 256 % (C/8) == 0, so a thread always sees the same 8 channels
  const int cg = threadIdx.x % (C >> 3);
  float sc[8], sh[8], rsc[8], rsh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j];
    rsc[j] = RES_AFFINE ? res_scale[cg * 8 + j] : 1.f; rsh[j] = RES_AFFINE ? res_shift[cg * 8 + j] : 0.f;
  }
  // FLAT_APPLY consecutive 256-vector rows per block, one vector of each per thread, no loop over the tensor
  // (tests/probes/probe_stream.hip: 6.1-6.4 TB/s; a grid of 8192 blocks striding the tensor: 5.2)
  const int64_t base = (int64_t)blockIdx.x * (256 * FLAT_APPLY) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < FLAT_APPLY; ++u) {
    const int64_t i = base + u * 256;
    if (i >= nvec) break;
    float f[8], rsd[8];
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + i * 8)), f);
    if (residual) unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(residual + i * 8)), rsd);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = f[j] * sc[j] + sh[j];
      // res_scale / res_shift: the residual is a RAW convolution output (a downsample block's 1x1 shortcut) normalised here,
      // rounded to bf16 exactly as the pass that used to materialise it did -- bit-identical, one write + one read less
      if (residual) v += RES_AFFINE ? bf16_bits_to_f32(f32_to_bf16_bits(rsd[j] * rsc[j] + rsh[j])) : rsd[j];
      if (relu) v = fmaxf(v, 0.f);
      f[j] = v;
    }
    if (relu_mask) {                                      // bit j: output j of this 8-channel vector is positive
      unsigned m = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) m |= (f[j] > 0.f ? 1u : 0u) << j;
      relu_mask[i] = (unsigned char)m;
    }
    // streaming store: the tensor is far larger than L2 / MALL and is not read again by this kernel
    __builtin_nontemporal_store(pack8(f), reinterpret_cast<u32x4*>(y + i * 8));
  }
}

// gather form: each input pixel looks at the <= 2x2 outputs whose window covers it.  Returns the gradient of the
// 8 channels cg*8.. of input pixel (n, hi, wi), rounded to bf16 like the materialised tensor would be.
__device__ __forceinline__ void pooled_grad8(const unsigned char* __restrict__ argmax, const unsigned short* __restrict__ dy,
                                             const PoolGeom& g_, int n, int hi, int wi, int cg, float (&out)[8]) {
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  // outputs ho with ho*2-1 <= hi <= ho*2+1  ->  ho in [ceil((hi-1)/2), floor((hi+1)/2)]
  const int ho_lo = hi >> 1, ho_hi = (hi + 1) >> 1;
  const int wo_lo = (wi) >> 1, wo_hi = (wi + 1) >> 1;
  for (int ho = ho_lo; ho <= ho_hi; ++ho) {
    if (ho >= g_.Ho) continue;
    const int kh = hi - (ho * 2 - 1);
    if (kh < 0 || kh > 2) continue;
    for (int wo = wo_lo; wo <= wo_hi; ++wo) {
      if (wo >= g_.Wo) continue;
      const int kw = wi - (wo * 2 - 1);
      if (kw < 0 || kw > 2) continue;
      const int64_t o = (((int64_t)n * g_.Ho + ho) * g_.Wo + wo) * g_.C + cg * 8;
      const u32x2 am = *reinterpret_cast<const u32x2*>(argmax + o);
      float g[8];
      unpack8(*reinterpret_cast<const u32x4*>(dy + o), g);
      const unsigned code = (unsigned)(kh * 3 + kw);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned b = (am[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
        if (b == code) acc[j] += g[j];
      }
    }
  }
  unpack8(pack8(acc), out);
}

// MODE: where the ReLU mask comes from -- 0 no ReLU, 1 the output y, 2 recomputed from x (scale, shift), 3 mask bits.
// Compile-time so that each variant only carries the registers it needs (occupancy is what hides the HBM latency here).
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const unsigned short* __restrict__ dy,
                                                             const unsigned short* __restrict__ x,
                                                             const unsigned short* __restrict__ y,
                                                             const unsigned char* __restrict__ relu_mask,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int64_t rows, int C,
                                                             int relu, const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             double* __restrict__ dgamma,
                                                             double* __restrict__ dbeta) {
  const int tid = threadIdx.x, cgs = C >> 3, cg = tid % cgs, rl = tid / cgs, rls = 256 / cgs;
  float acc[2][8], mu[8], rs[8], sc[8], sh[8];
  constexpr bool from_x = MODE == 2;
  (void)relu;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc[0][j] = 0.f; acc[1][j] = 0.f; mu[j] = mean[cg * 8 + j]; rs[j] = rstd[cg * 8 + j];
    sc[j] = from_x ? scale[cg * 8 + j] : 0.f; sh[j] = from_x ? shift[cg * 8 + j] : 0.f;
  }
  // a block sums a CONTIGUOUS range of rows (not a grid-strided set): DRAM pages are walked in order
  const int64_t per_block = ((rows + gridDim.x - 1) / gridDim.x + rls - 1) / rls * rls;
  const int64_t r_end = min(rows, ((int64_t)blockIdx.x + 1) * per_block);
#pragma unroll 2
  for (int64_t r = (int64_t)blockIdx.x * per_block + rl; r < r_end; r += rls) {
    float g[8], xv[8], yv[8];
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(dy + r * C + cg * 8)), g);
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + r * C + cg * 8)), xv);
    if (from_x) {
#pragma unroll
      for (int j = 0; j < 8; ++j) yv[j] = xv[j] * sc[j] + sh[j];
    } else if (MODE == 3) {                                // 1 bit per element instead of re-reading the output
      const unsigned m = relu_mask[r * (C >> 3) + cg];
#pragma unroll
      for (int j = 0; j < 8; ++j) yv[j] = ((m >> j) & 1u) ? 1.f : 0.f;
    } else if (MODE == 1) unpack8(*reinterpret_cast<const u32x4*>(y + r * C + cg * 8), yv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float dz = (MODE != 0 && !(yv[j] > 0.f)) ? 0.f : g[j];
      acc[0][j] += dz * ((xv[j] - mu[j]) * rs[j]);
      acc[1][j] += dz;
    }
  }
  double* const dst[2] = {dgamma, dbeta};
  block_reduce_to_global<2>(acc, C, dst);
}

template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x, const unsigned short* __restrict__ y,
    const unsigned char* __restrict__ relu_mask,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
    const double* __restrict__ dgamma, const double* __restrict__ dbeta, int64_t rows, int C, int relu,
    const float* __restrict__ scale, const float* __restrict__ shift,
    unsigned short* __restrict__ dx, unsigned short* __restrict__ d_residual, float* __restrict__ dgamma_f32,
    float* __restrict__ dbeta_f32) {
  const int cgs = C >> 3;
  constexpr bool from_x = MODE == 2;
  (void)relu;
  const int64_t nvec = rows * cgs;
  const float inv_rows = 1.f / (float)rows;
  if (blockIdx.x == 0 && dgamma_f32) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      dgamma_f32[c] += (float)dgamma[c];
      dbeta_f32[c] += (float)dbeta[c];
    }
  }
  // per-thread channel constants (a block starts at a multiple of 256 vectors, 256 % (C/8) == 0): dx = k1*dz - k2 - xh*k3
  const int cg = threadIdx.x % cgs;
  // dx = k1*(dz - k2 - xh*k3), xh = (x - mu)*rs  ==  kA*dz + (kB*x + kD): three constants and two FMAs per element
  float kA[8], kB[8], kD[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    const float mu = mean[c], rs = rstd[c], k1 = gamma[c] * rs;
    const float k2 = (float)dbeta[c] * inv_rows, k3 = (float)dgamma[c] * inv_rows;
    kA[j] = k1; kB[j] = -k1 * k3 * rs; kD[j] = k1 * (k3 * rs * mu - k2);
    sc[j] = from_x ? scale[c] : 0.f; sh[j] = from_x ? shift[c] : 0.f;
  }
  const int64_t base = (int64_t)blockIdx.x * (256 * FLAT_BWD) + threadIdx.x;     // flat: see bn_apply_kernel
#pragma unroll
  for (int u = 0; u < FLAT_BWD; ++u) {
    const int64_t i = base + u * 256;
    if (i >= nvec) break;
    float g[8], xv[8], yv[8], o[8];
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(dy + i * 8)), g);
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + i * 8)), xv);
    if (from_x) {
#pragma unroll
      for (int j = 0; j < 8; ++j) yv[j] = xv[j] * sc[j] + sh[j];
    } else if (MODE == 3) {
      const unsigned m = relu_mask[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) yv[j] = ((m >> j) & 1u) ? 1.f : 0.f;
    } else if (MODE == 1) unpack8(*reinterpret_cast<const u32x4*>(y + i * 8), yv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float dz = (MODE != 0 && !(yv[j] > 0.f)) ? 0.f : g[j];
      g[j] = dz;
      o[j] = kA[j] * dz + (kB[j] * xv[j] + kD[j]);
    }
    // streaming stores: the tensors are far larger than L2 / MALL and are not read again by this kernel
    __builtin_nontemporal_store(pack8(o), reinterpret_cast<u32x4*>(dx + i * 8));
    if (d_residual) __builtin_nontemporal_store(pack8(g), reinterpret_cast<u32x4*>(d_residual + i * 8));
  }
}

// ---------------------------------------------------------------- stem: BatchNorm backward fed by the pooled gradient
// The gradient of the stem activation is the 3x3/2 max-pool backward of the pooled gradient gp through argmax; it is
// not materialised.  A thread owns the 2x2 input pixels (2a+dy, 2b+dx) of one 8-channel group: they are covered by the
// pooling windows (a+i, b+j), i, j in {0, 1} only -- window (a, b) covers all four, (a, b+1) the right column,
// (a+1, b) the lower row, (a+1, b+1) the lower right pixel -- so four (gradient, argmax) loads serve four pixels.
// Contributions are summed in the order of the materialising kernel and rounded to bf16 like its output.
__device__ __forceinline__ void pooled_grad_2x2(const unsigned char* __restrict__ argmax, const unsigned short* __restrict__ gp,
                                                const PoolGeom& g_, int n, int a, int b, int cg, float (&out)[4][8]) {
  isic_pool::Windows w;
  isic_pool::load_windows(w, argmax, gp, g_, n, a, b, cg);
  isic_pool::windows_to_grad(w, g_, a, b, out);
}

__global__ __launch_bounds__(256) void stem_bn_bwd_reduce_kernel(const unsigned char* __restrict__ argmax,
                                                                  const unsigned short* __restrict__ gp, PoolGeom geom,
                                                                  const unsigned short* __restrict__ x,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, int N,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  double* __restrict__ dgamma, double* __restrict__ dbeta) {
  const int C = geom.C, tid = threadIdx.x, cgs = C >> 3, cg = tid % cgs, rl = tid / cgs, rls = 256 / cgs;
  float acc[2][8], mu[8], rs[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc[0][j] = 0.f; acc[1][j] = 0.f; mu[j] = mean[cg * 8 + j]; rs[j] = rstd[cg * 8 + j];
    sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j];
  }
  const int Hb = (geom.H + 1) >> 1, Wb = (geom.W + 1) >> 1;
  const int64_t nblk = (int64_t)N * Hb * Wb;
  for (int64_t q = (int64_t)blockIdx.x * rls + rl; q < nblk; q += (int64_t)gridDim.x * rls) {
    const int b = (int)(q % Wb);
    const int64_t t = q / Wb;
    const int a = (int)(t % Hb), n = (int)(t / Hb);
    u32x4 xr[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {                       // clamped addresses: every load of the thread in flight at once
      const int hi = min(2 * a + (p >> 1), geom.H - 1), wi = min(2 * b + (p & 1), geom.W - 1);
      xr[p] = *reinterpret_cast<const u32x4*>(x + (((int64_t)n * geom.H + hi) * geom.W + wi) * C + cg * 8);
    }
    float g[4][8];
    pooled_grad_2x2(argmax, gp, geom, n, a, b, cg, g);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const bool in = (2 * a + (p >> 1) < geom.H) && (2 * b + (p & 1) < geom.W);
      float xv[8];
      unpack8(xr[p], xv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dz = (in && xv[j] * sc[j] + sh[j] > 0.f) ? g[p][j] : 0.f;
        acc[0][j] += dz * ((xv[j] - mu[j]) * rs[j]);
        acc[1][j] += dz;
      }
    }
  }
  double* const dst[2] = {dgamma, dbeta};
  block_reduce_to_global<2>(acc, C, dst);
}

__global__ __launch_bounds__(256) void stem_bn_bwd_apply_kernel(
    const unsigned char* __restrict__ argmax, const unsigned short* __restrict__ gp, PoolGeom geom,
    const unsigned short* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const double* __restrict__ dgamma, const double* __restrict__ dbeta, int N,
    const float* __restrict__ scale, const float* __restrict__ shift, unsigned short* __restrict__ dx,
    float* __restrict__ dgamma_f32, float* __restrict__ dbeta_f32) {
  const int C = geom.C, cgs = C >> 3;
  const float inv_rows = 1.f / (float)((int64_t)N * geom.H * geom.W);
  if (blockIdx.x == 0 && dgamma_f32) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      dgamma_f32[c] += (float)dgamma[c];
      dbeta_f32[c] += (float)dbeta[c];
    }
  }
  const int cg = threadIdx.x % cgs;           // (the grid stride is a multiple of C/8)
  float kA[8], kB[8], kD[8], sc[8], sh[8];              // as in bn_bwd_apply_kernel
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    const float mu = mean[c], rs = rstd[c], k1 = gamma[c] * rs;
    const float k2 = (float)dbeta[c] * inv_rows, k3 = (float)dgamma[c] * inv_rows;
    kA[j] = k1; kB[j] = -k1 * k3 * rs; kD[j] = k1 * (k3 * rs * mu - k2);
    sc[j] = scale[c]; sh[j] = shift[c];
  }
  const int Hb = (geom.H + 1) >> 1, Wb = (geom.W + 1) >> 1;
  const int64_t nvec = (int64_t)N * Hb * Wb * cgs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = i / cgs;
    const int b = (int)(q % Wb);
    const int64_t t = q / Wb;
    const int a = (int)(t % Hb), n = (int)(t / Hb);
    u32x4 xr[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {                       // clamped addresses: every load of the thread in flight at once
      const int hi = min(2 * a + (p >> 1), geom.H - 1), wi = min(2 * b + (p & 1), geom.W - 1);
      xr[p] = *reinterpret_cast<const u32x4*>(x + (((int64_t)n * geom.H + hi) * geom.W + wi) * C + cg * 8);
    }
    float g[4][8];
    pooled_grad_2x2(argmax, gp, geom, n, a, b, cg, g);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int hi = 2 * a + (p >> 1), wi = 2 * b + (p & 1);
      if (hi >= geom.H || wi >= geom.W) continue;
      const int64_t off = (((int64_t)n * geom.H + hi) * geom.W + wi) * C + cg * 8;
      float xv[8], o[8];
      unpack8(xr[p], xv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dz = (xv[j] * sc[j] + sh[j] > 0.f) ? g[p][j] : 0.f;
        o[j] = kA[j] * dz + (kB[j] * xv[j] + kD[j]);
      }
      __builtin_nontemporal_store(pack8(o), reinterpret_cast<u32x4*>(dx + off));
    }
  }
}

// ---------------------------------------------------------------- pooling
// scale != nullptr: the input is a raw convolution output and y = maxpool(relu(x * scale + shift)) -- the BatchNorm
// apply and ReLU of the stem are done on the fly (each value rounded to bf16 as the materialised tensor would be)
// A thread makes TWO horizontally adjacent outputs (wo, wo + 1) of one 8-channel group from the 3 x 5 input pixels they
// cover -- 7.5 instead of 9 loads per output (the kernel is bound by its L2 reads: neighbouring windows overlap) -- all at
// clamped, always valid addresses so that the fifteen loads are in flight together (1.51 -> 1.44 ms at 2048 images; capping
// it at 128 VGPRs for a fourth wave per SIMD spills and runs 2.3 ms).
template <bool AFFINE>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const unsigned short* __restrict__ x,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           unsigned short* __restrict__ y,
                                                           unsigned char* __restrict__ argmax,
                                                           unsigned short* __restrict__ xsel, int N, int H, int W,
                                                           int C, int Ho, int Wo) {
  const int cgs = C >> 3, Wo2 = (Wo + 1) >> 1;
  const int64_t nvec = (int64_t)N * Ho * Wo2 * cgs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cgs);
    int64_t t = i / cgs;
    const int wp = (int)(t % Wo2); t /= Wo2;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int wo0 = wp * 2;
    u32x4 raw[3][5];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = min(max(ho * 2 - 1 + kh, 0), H - 1);
#pragma unroll
      for (int c5 = 0; c5 < 5; ++c5) {
        const int wi = min(max(wo0 * 2 - 1 + c5, 0), W - 1);
        raw[kh][c5] = *reinterpret_cast<const u32x4*>(x + (((int64_t)n * H + hi) * W + wi) * C + cg * 8);
      }
    }
    float sc[8], sh[8];
    if (AFFINE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j]; }
    }
    // Running maxima of the two outputs: the value, and (raw bf16 bits << 16 | tap code) of the tap it came from -- ONE
    // select moves both.  Every input pixel is normalised once and offered to the windows it belongs to in scan order
    // (first maximum wins); the validity test is per pixel, the eight channel updates under it are selects, not branches
    // (the compiler made `if (in && f > best) {...}` 144 exec-mask regions with a branch each: 2.5 -> 2.0 ms at 4096 images).
    float best[2][8];
    unsigned pk[2][8];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int j = 0; j < 8; ++j) { best[o][j] = -INFINITY; pk[o][j] = 0u; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = ho * 2 - 1 + kh;
#pragma unroll
      for (int c5 = 0; c5 < 5; ++c5) {
        const int wi = wo0 * 2 - 1 + c5;
        const bool in = hi >= 0 && hi < H && wi >= 0 && wi < W;
        unsigned rb[8];
        float f[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) { rb[2 * q] = raw[kh][c5][q] << 16; rb[2 * q + 1] = raw[kh][c5][q] & 0xFFFF0000u; }
        if (AFFINE) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaxf(__uint_as_float(rb[j]) * sc[j] + sh[j], 0.f);
          unpack8(pack8(f), f);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = __uint_as_float(rb[j]);
        }
        if (in) {
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            const int kw = c5 - 2 * o;                   // this pixel's column inside window o
            if (kw < 0 || kw > 2) continue;
            const unsigned code = (unsigned)(kh * 3 + kw);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const bool take = f[j] > best[o][j];       // first maximum wins
              best[o][j] = take ? f[j] : best[o][j];
              pk[o][j] = take ? (rb[j] | code) : pk[o][j];
            }
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int wo = wo0 + o;
      if (wo >= Wo) break;
      const int64_t oi = ((((int64_t)n * Ho + ho) * Wo + wo) * cgs + cg);
      __builtin_nontemporal_store(pack8(best[o]), reinterpret_cast<u32x4*>(y + oi * 8));
      if (xsel) {
        u32x4 xs;
#pragma unroll
        for (int q = 0; q < 4; ++q) xs[q] = (pk[o][2 * q] >> 16) | (pk[o][2 * q + 1] & 0xFFFF0000u);
        __builtin_nontemporal_store(xs, reinterpret_cast<u32x4*>(xsel + oi * 8));
      }
      if (argmax) {
        u32x2 p;
        p[0] = (pk[o][0] & 0xFFu) | ((pk[o][1] & 0xFFu) << 8) | ((pk[o][2] & 0xFFu) << 16) | (pk[o][3] << 24);
        p[1] = (pk[o][4] & 0xFFu) | ((pk[o][5] & 0xFFu) << 8) | ((pk[o][6] & 0xFFu) << 16) | (pk[o][7] << 24);
        __builtin_nontemporal_store(p, reinterpret_cast<u32x2*>(argmax + oi * 8));
      }
    }
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned char* __restrict__ argmax,
                                                           const unsigned short* __restrict__ dy,
                                                           unsigned short* __restrict__ dx, int N, int H, int W, int C,
                                                           int Ho, int Wo) {
  const int cgs = C >> 3;
  const PoolGeom geom{H, W, C, Ho, Wo};
  const int64_t nvec = (int64_t)N * H * W * cgs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cgs);
    int64_t t = i / cgs;
    const int wi = (int)(t % W); t /= W;
    const int hi = (int)(t % H);
    const int n = (int)(t / H);
    float acc[8];
    pooled_grad8(argmax, dy, geom, n, hi, wi, cg, acc);
    *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(acc);
  }
}

// one block per (image, 64-channel group): 4 waves stride the pixels
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const unsigned short* __restrict__ x, float* __restrict__ y,
                                                           int HW, int C) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x, c = blockIdx.y * 64 + lane;
  float s = 0.f;
  if (c < C)
    for (int p = wave; p < HW; p += 4) s += bf16_bits_to_f32(x[((int64_t)n * HW + p) * C + c]);
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < C)
    y[(int64_t)n * C + c] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / (float)HW;
}

__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, unsigned short* __restrict__ dx, int64_t total,
                                   int HW, int C) {
  const float inv = 1.f / (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t n = i / ((int64_t)HW * C);
    dx[i] = f32_to_bf16_bits(dy[n * C + c] * inv);
  }
}

__global__ void add_bf16_kernel(unsigned short* __restrict__ a, const unsigned short* __restrict__ b, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    float fa[8], fb[8];
    unpack8(*reinterpret_cast<const u32x4*>(a + i * 8), fa);
    unpack8(*reinterpret_cast<const u32x4*>(b + i * 8), fb);
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] += fb[j];
    *reinterpret_cast<u32x4*>(a + i * 8) = pack8(fa);
  }
}

// NCHW (fp32 or bf16) -> NHWC with C padded to 4, bf16.  One thread per output pixel.
__global__ void nchw_to_nhwc4_kernel(const void* __restrict__ in, int in_is_bf16, unsigned short* __restrict__ out,
                                     int64_t npix_total, int C, int64_t HW) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix_total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / HW, p = i - n * HW;
    unsigned short v[4] = {0, 0, 0, 0};
    for (int c = 0; c < C && c < 4; ++c) {
      const int64_t src = (n * C + c) * HW + p;
      v[c] = in_is_bf16 ? reinterpret_cast<const unsigned short*>(in)[src]
                        : f32_to_bf16_bits(reinterpret_cast<const float*>(in)[src]);
    }
    u32x2 o;
    o[0] = v[0] | ((unsigned)v[1] << 16);
    o[1] = v[2] | ((unsigned)v[3] << 16);
    *reinterpret_cast<u32x2*>(out + i * 4) = o;
  }
}

// The same for HW % 4 == 0: four pixels per thread (8-byte / 16-byte plane loads, two 16-byte stores), image = blockIdx.y
// (no 64-bit division per pixel).  0.46 -> 0.3 ms for 2048 images of 224 x 224.
template <bool BF16>
__global__ __launch_bounds__(256) void nchw_to_nhwc4_x4_kernel(const void* __restrict__ in, unsigned short* __restrict__ out,
                                                                int N, int C, int HW) {
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= HW) return;
  for (int n = blockIdx.y; n < N; n += gridDim.y) {
    unsigned short v[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < C) {
        const size_t src = ((size_t)n * C + c) * HW + p;
        if (BF16) {
          const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(in) + src));
          v[0][c] = (unsigned short)(w[0] & 0xFFFFu); v[1][c] = (unsigned short)(w[0] >> 16);
          v[2][c] = (unsigned short)(w[1] & 0xFFFFu); v[3][c] = (unsigned short)(w[1] >> 16);
        } else {
          const f32x4 w = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(in) + src));
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q][c] = f32_to_bf16_bits(w[q]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q][c] = 0;
      }
    }
    u32x4 o0, o1;
    o0[0] = v[0][0] | ((unsigned)v[0][1] << 16); o0[1] = v[0][2] | ((unsigned)v[0][3] << 16);
    o0[2] = v[1][0] | ((unsigned)v[1][1] << 16); o0[3] = v[1][2] | ((unsigned)v[1][3] << 16);
    o1[0] = v[2][0] | ((unsigned)v[2][1] << 16); o1[1] = v[2][2] | ((unsigned)v[2][3] << 16);
    o1[2] = v[3][0] | ((unsigned)v[3][1] << 16); o1[3] = v[3][2] | ((unsigned)v[3][3] << 16);
    unsigned short* dst = out + ((size_t)n * HW + p) * 4;
    __builtin_nontemporal_store(o0, reinterpret_cast<u32x4*>(dst));
    __builtin_nontemporal_store(o1, reinterpret_cast<u32x4*>(dst + 8));
  }
}

// Streaming passes: one 16-byte vector per thread and as many blocks as that takes.  tests/probes/probe_stream.hip, 822 MB in
// + 822 MB out: 6.4 TB/s that way against 5.2 TB/s for 8192 blocks striding the tensor (the launch shape these kernels had).
constexpr int STREAM_CAP = 1 << 22;
inline int grid_for(int64_t n, int block, int cap = 8192) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
inline bool bn_c_ok(int C) { return C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0; }

}  // namespace

extern "C" {

int isic_bn_stats_bf16(const uint16_t* x, int64_t rows, int C, double* sum, double* sumsq, void* stream) {
  ISIC_CHECK_ARG(x && sum && sumsq && rows > 0 && C > 0);
  if (!bn_c_ok(C)) return ISIC_ERR_UNSUPPORTED;
  const int rls = 256 / (C / 8);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(grid_for(rows, rls * 8, 2048)), dim3(256), 0, as_stream(stream), x, rows, C,
                     sum, sumsq);
  return isic_launch_status();
}

int isic_bn_finalize(const double* sum, const double* sumsq, int nslots, int64_t rows, int C, const float* gamma,
                     const float* beta, float eps, float momentum, float* scale, float* shift, float* mean,
                     float* rstd, float* running_mean, float* running_var, void* stream) {
  ISIC_CHECK_ARG(sum && sumsq && gamma && beta && scale && shift && mean && rstd && rows > 0 && C > 0 && nslots > 0);
  ISIC_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, as_stream(stream), sum, sumsq, nslots,
                     rows, C,
                     gamma, beta, eps, momentum, scale, shift, mean, rstd, running_mean, running_var);
  return isic_launch_status();
}

int isic_bn_bwd_finalize(const double* sum_dz, const double* sum_dzy, int nslots, int C, const float* mean,
                         const float* rstd, double* dgamma, double* dbeta, void* stream) {
  ISIC_CHECK_ARG(sum_dz && sum_dzy && mean && rstd && dgamma && dbeta && nslots > 0 && C > 0);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, as_stream(stream), sum_dz, sum_dzy, nslots,
                     C, mean, rstd, dgamma, dbeta);
  return isic_launch_status();
}

int isic_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, int C, float* scale, float* shift, void* stream) {
  ISIC_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && C > 0);
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, as_stream(stream), gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  return isic_launch_status();
}

int isic_bn_apply_bf16(const uint16_t* x, const float* scale, const float* shift, const uint16_t* residual,
                       uint16_t* y, int64_t rows, int C, int relu, void* stream) {
  ISIC_CHECK_ARG(x && scale && shift && y && rows > 0 && C > 0 && C % 8 == 0);
  if (!bn_c_ok(C)) return ISIC_ERR_UNSUPPORTED;   // per-thread channel constants: C/8 must divide the 256-thread block
  const int64_t nvec = rows * (C / 8);
  hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid_for(nvec, 256 * FLAT_APPLY, STREAM_CAP)), dim3(256), 0, as_stream(stream), x, scale, shift,
                     residual, y, nullptr, nvec, C, relu, nullptr, nullptr);
  return isic_launch_status();
}

int isic_bn_apply_mask_bf16(const uint16_t* x, const float* scale, const float* shift, const uint16_t* residual,
                            uint16_t* y, uint8_t* relu_mask, int64_t rows, int C, void* stream) {
  ISIC_CHECK_ARG(x && scale && shift && y && relu_mask && rows > 0 && C > 0 && C % 8 == 0);
  if (!bn_c_ok(C)) return ISIC_ERR_UNSUPPORTED;   // per-thread channel constants: C/8 must divide the 256-thread block
  const int64_t nvec = rows * (C / 8);
  hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid_for(nvec, 256 * FLAT_APPLY, STREAM_CAP)), dim3(256), 0, as_stream(stream), x, scale, shift,
                     residual, y, relu_mask, nvec, C, 1, nullptr, nullptr);
  return isic_launch_status();
}

int isic_bn_apply_mask_res_affine_bf16(const uint16_t* x, const float* scale, const float* shift, const uint16_t* residual_raw,
                                       const float* res_scale, const float* res_shift, uint16_t* y, uint8_t* relu_mask,
                                       int64_t rows, int C, void* stream) {
  ISIC_CHECK_ARG(x && scale && shift && residual_raw && res_scale && res_shift && y && relu_mask && rows > 0 && C > 0 && C % 8 == 0);
  if (!bn_c_ok(C)) return ISIC_ERR_UNSUPPORTED;
  const int64_t nvec = rows * (C / 8);
  hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(grid_for(nvec, 256 * FLAT_APPLY, STREAM_CAP)), dim3(256), 0, as_stream(stream), x, scale, shift,
                     residual_raw, y, relu_mask, nvec, C, 1, res_scale, res_shift);
  return isic_launch_status();
}

int isic_bn_bwd_reduce_bf16(const uint16_t* dy, const uint16_t* x, const uint16_t* y, const float* mean,
                            const float* rstd, int64_t rows, int C, int relu, const float* scale,
                            const float* shift, double* dgamma, double* dbeta, void* stream) {
  ISIC_CHECK_ARG(dy && x && mean && rstd && dgamma && dbeta && rows > 0 && C > 0);
  ISIC_CHECK_ARG((scale == nullptr) == (shift == nullptr));
  ISIC_CHECK_ARG(!relu || y || scale);
  if (!bn_c_ok(C)) return ISIC_ERR_UNSUPPORTED;
  const int rls = 256 / (C / 8);
  const dim3 grid(grid_for(rows, rls * 8, 2048));
#define ISIC_REDUCE(M) hipLaunchKernelGGL(bn_bwd_reduce_kernel<M>, grid, dim3(256), 0, as_stream(stream), dy, x, y, nullptr, \
                                          mean, rstd, rows, C, relu, scale, shift, dgamma, dbeta)
  if (!relu) ISIC_REDUCE(0); else if (scale) ISIC_REDUCE(2); else ISIC_REDUCE(1);
#undef ISIC_REDUCE
  return isic_launch_status();
}

int isic_bn_bwd_reduce_mask_bf16(const uint16_t* dy, const uint16_t* x, const uint8_t* relu_mask, const float* mean,
                                 const float* rstd, int64_t rows, int C, double* dgamma, double* dbeta, void* stream) {
  ISIC_CHECK_ARG(dy && x && relu_mask && mean && rstd && dgamma && dbeta && rows > 0 && C > 0);
  if (!bn_c_ok(C)) return ISIC_ERR_UNSUPPORTED;
  const int rls = 256 / (C / 8);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel<3>, dim3(grid_for(rows, rls * 8, 2048)), dim3(256), 0, as_stream(stream), dy, x,
                     nullptr, relu_mask, mean, rstd, rows, C, 1, nullptr, nullptr, dgamma, dbeta);
  return isic_launch_status();
}

int isic_bn_bwd_reduce_pooled_bf16(const uint8_t* argmax, const uint16_t* dy_pooled, const uint16_t* x, const float* mean,
                                   const float* rstd, int N, int H, int W, int C, int Ho, int Wo, const float* scale,
                                   const float* shift, double* dgamma, double* dbeta, void* stream) {
  ISIC_CHECK_ARG(argmax && dy_pooled && x && mean && rstd && scale && shift && dgamma && dbeta);
  ISIC_CHECK_ARG(N > 0 && H > 0 && W > 0 && Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1);
  if (!bn_c_ok(C)) return ISIC_ERR_UNSUPPORTED;
  const int64_t nblk = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2);
  const int rls = 256 / (C / 8);
  hipLaunchKernelGGL(stem_bn_bwd_reduce_kernel, dim3(grid_for(nblk, rls * 2, 2048)), dim3(256), 0, as_stream(stream),
                     argmax, dy_pooled, PoolGeom{H, W, C, Ho, Wo}, x, mean, rstd, N, scale, shift, dgamma, dbeta);
  return isic_launch_status();
}

int isic_bn_bwd_apply_bf16(const uint16_t* dy, const uint16_t* x, const uint16_t* y, const float* mean,
                           const float* rstd, const float* gamma, const double* dgamma, const double* dbeta,
                           int64_t rows, int C, int relu, const float* scale, const float* shift, uint16_t* dx,
                           uint16_t* d_residual, float* dgamma_f32, float* dbeta_f32, void* stream) {
  ISIC_CHECK_ARG(dy && x && mean && rstd && gamma && dgamma && dbeta && dx && rows > 0 && C > 0 && C % 8 == 0);
  ISIC_CHECK_ARG((scale == nullptr) == (shift == nullptr));
  ISIC_CHECK_ARG(!relu || y || scale);
  ISIC_CHECK_ARG((dgamma_f32 == nullptr) == (dbeta_f32 == nullptr));
  if (!bn_c_ok(C)) return ISIC_ERR_UNSUPPORTED;   // per-thread channel constants: C/8 must divide the 256-thread block
  const int64_t nvec = rows * (C / 8);
  const dim3 grid(grid_for(nvec, 256 * FLAT_BWD, STREAM_CAP));
#define ISIC_APPLY(M) hipLaunchKernelGGL(bn_bwd_apply_kernel<M>, grid, dim3(256), 0, as_stream(stream), dy, x, y, nullptr, mean, \
                                         rstd, gamma, dgamma, dbeta, rows, C, relu, scale, shift, dx, d_residual, dgamma_f32, dbeta_f32)
  if (!relu) ISIC_APPLY(0); else if (scale) ISIC_APPLY(2); else ISIC_APPLY(1);
#undef ISIC_APPLY
  return isic_launch_status();
}

int isic_bn_bwd_apply_mask_bf16(const uint16_t* dy, const uint16_t* x, const uint8_t* relu_mask, const float* mean,
                                const float* rstd, const float* gamma, const double* dgamma, const double* dbeta,
                                int64_t rows, int C, uint16_t* dx, uint16_t* d_residual, float* dgamma_f32,
                                float* dbeta_f32, void* stream) {
  ISIC_CHECK_ARG(dy && x && relu_mask && mean && rstd && gamma && dgamma && dbeta && dx && rows > 0 && C > 0 && C % 8 == 0);
  ISIC_CHECK_ARG((dgamma_f32 == nullptr) == (dbeta_f32 == nullptr));
  if (!bn_c_ok(C)) return ISIC_ERR_UNSUPPORTED;   // per-thread channel constants: C/8 must divide the 256-thread block
  const int64_t nvec = rows * (C / 8);
  hipLaunchKernelGGL(bn_bwd_apply_kernel<3>, dim3(grid_for(nvec, 256 * FLAT_BWD, STREAM_CAP)), dim3(256), 0, as_stream(stream), dy, x, nullptr,
                     relu_mask, mean, rstd, gamma, dgamma, dbeta, rows, C, 1, nullptr, nullptr, dx, d_residual, dgamma_f32,
                     dbeta_f32);
  return isic_launch_status();
}

int isic_bn_bwd_apply_pooled_bf16(const uint8_t* argmax, const uint16_t* dy_pooled, const uint16_t* x, const float* mean,
                                  const float* rstd, const float* gamma, const double* dgamma, const double* dbeta, int N,
                                  int H, int W, int C, int Ho, int Wo, const float* scale, const float* shift,
                                  uint16_t* dx, float* dgamma_f32, float* dbeta_f32, void* stream) {
  ISIC_CHECK_ARG(argmax && dy_pooled && x && mean && rstd && gamma && dgamma && dbeta && scale && shift && dx);
  ISIC_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1);
  ISIC_CHECK_ARG((dgamma_f32 == nullptr) == (dbeta_f32 == nullptr));
  if (!bn_c_ok(C)) return ISIC_ERR_UNSUPPORTED;
  const int64_t nvec = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
  hipLaunchKernelGGL(stem_bn_bwd_apply_kernel, dim3(grid_for(nvec, 256)), dim3(256), 0, as_stream(stream), argmax,
                     dy_pooled, PoolGeom{H, W, C, Ho, Wo}, x, mean, rstd, gamma, dgamma, dbeta, N, scale, shift, dx,
                     dgamma_f32, dbeta_f32);
  return isic_launch_status();
}

int isic_maxpool3x3s2_fwd_bf16(const uint16_t* x, uint16_t* y, uint8_t* argmax, int N, int H, int W, int C, int Ho,
                               int Wo, void* stream) {
  ISIC_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
  ISIC_CHECK_ARG(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1);
  const int64_t nvec = (int64_t)N * Ho * ((Wo + 1) / 2) * (C / 8);           // a thread makes two adjacent outputs
  hipLaunchKernelGGL(maxpool_fwd_kernel<false>, dim3(grid_for(nvec, 256, STREAM_CAP)), dim3(256), 0, as_stream(stream), x, nullptr,
                     nullptr, y, argmax, nullptr, N, H, W, C, Ho, Wo);
  return isic_launch_status();
}

int isic_bn_relu_maxpool3x3s2_fwd_bf16(const uint16_t* x, const float* scale, const float* shift, uint16_t* y,
                                       uint8_t* argmax, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
  return isic_bn_relu_maxpool3x3s2_fwd_sel_bf16(x, scale, shift, y, argmax, nullptr, N, H, W, C, Ho, Wo, stream);
}

int isic_bn_relu_maxpool3x3s2_fwd_sel_bf16(const uint16_t* x, const float* scale, const float* shift, uint16_t* y,
                                           uint8_t* argmax, uint16_t* x_sel, int N, int H, int W, int C, int Ho, int Wo,
                                           void* stream) {
  ISIC_CHECK_ARG(x && scale && shift && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
  ISIC_CHECK_ARG(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1);
  const int64_t nvec = (int64_t)N * Ho * ((Wo + 1) / 2) * (C / 8);           // a thread makes two adjacent outputs
  hipLaunchKernelGGL(maxpool_fwd_kernel<true>, dim3(grid_for(nvec, 256, STREAM_CAP)), dim3(256), 0, as_stream(stream), x, scale,
                     shift, y, argmax, x_sel, N, H, W, C, Ho, Wo);
  return isic_launch_status();
}

int isic_maxpool3x3s2_bwd_bf16(const uint8_t* argmax, const uint16_t* dy, uint16_t* dx, int N, int H, int W, int C,
                               int Ho, int Wo, void* stream) {
  ISIC_CHECK_ARG(argmax && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
  const int64_t nvec = (int64_t)N * H * W * (C / 8);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(nvec, 256)), dim3(256), 0, as_stream(stream), argmax, dy, dx, N,
                     H, W, C, Ho, Wo);
  return isic_launch_status();
}

int isic_avgpool_fwd_bf16(const uint16_t* x, float* y, int N, int HW, int C, void* stream) {
  ISIC_CHECK_ARG(x && y && N > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(N, ceil_div(C, 64)), dim3(256), 0, as_stream(stream), x, y, HW, C);
  return isic_launch_status();
}

int isic_avgpool_bwd_bf16(const float* dy, uint16_t* dx, int N, int HW, int C, void* stream) {
  ISIC_CHECK_ARG(dy && dx && N > 0 && HW > 0 && C > 0);
  const int64_t total = (int64_t)N * HW * C;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), dy, dx, total, HW,
                     C);
  return isic_launch_status();
}

int isic_add_bf16(uint16_t* a, const uint16_t* b, int64_t n, void* stream) {
  ISIC_CHECK_ARG(a && b && n >= 0 && n % 8 == 0);
  if (n == 0) return ISIC_OK;
  hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, as_stream(stream), a, b, n / 8);
  return isic_launch_status();
}

int isic_nchw_to_nhwc4_bf16(const void* in, int in_is_bf16, uint16_t* out, int N, int C, int H, int W, void* stream) {
  ISIC_CHECK_ARG(in && out && N > 0 && C > 0 && C <= 4 && H > 0 && W > 0);
  const int64_t npix = (int64_t)N * H * W;
  const int64_t HW = (int64_t)H * W;
  if (HW % 4 == 0 && HW < (1 << 30) && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const dim3 grid((unsigned)((HW / 4 + 255) / 256), (unsigned)(N < 65535 ? N : 65535));
    if (in_is_bf16) hipLaunchKernelGGL(nchw_to_nhwc4_x4_kernel<true>, grid, dim3(256), 0, as_stream(stream), in, out, N, C, (int)HW);
    else hipLaunchKernelGGL(nchw_to_nhwc4_x4_kernel<false>, grid, dim3(256), 0, as_stream(stream), in, out, N, C, (int)HW);
    return isic_launch_status();
  }
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(grid_for(npix, 256, STREAM_CAP)), dim3(256), 0, as_stream(stream), in, in_is_bf16,
                     out, npix, C, HW);
  return isic_launch_status();
}

}  // extern "C"
