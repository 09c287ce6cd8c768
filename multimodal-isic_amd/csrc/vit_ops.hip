// ViT-S/16 companions of gemm_f16.hip (gfx950, fp16 activations, fp32 arithmetic inside): patch extraction, LayerNorm
// over 384-wide token rows, and multi-head self-attention over the 196 tokens of one image on MFMA.
//
// The reference's encoder is an un-vendored ConvMAE conv-ViT run frozen (save_latent.py:42-60); these implement the
// ViT-S/16 named by BASELINE.json configs[4] with timm semantics (pre-norm blocks, LayerNorm eps 1e-6, scaled dot
// product attention without a class token or masking: save_latent.py passes mask_ratio = 0).  oracle/vit.py is the fp32
// CPU restatement.

#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_h2(float lo, float hi) {
  const f16x2 h = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float h_lo(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[0]; }
__device__ __forceinline__ float h_hi(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[1]; }
__device__ __forceinline__ void unpack_h8(const u32x4 v, float (&f)[8]) {
  // (element by element through a scalar copy: bit-casting `v[i]` in place read element 0 four times with this hipcc)
  const unsigned w0 = v[0], w1 = v[1], w2 = v[2], w3 = v[3];
  f[0] = h_lo(w0); f[1] = h_hi(w0); f[2] = h_lo(w1); f[3] = h_hi(w1);
  f[4] = h_lo(w2); f[5] = h_hi(w2); f[6] = h_lo(w3); f[7] = h_hi(w3);
}
__device__ __forceinline__ u32x4 pack_h8(const float (&f)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_h2(f[2 * i], f[2 * i + 1]);
  return v;
}

// ---------------------------------------------------------------- patches
// images NCHW (fp32 or bf16/fp16 bits are not mixed here: fp32 in) -> rows [n*gh*gw + py*gw + px][c*P*P + ky*P + kx]
// fp16: the im2col of a PxP / stride P convolution (Conv2d weight [D][C][P][P] flattened is then a Linear weight).
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, unsigned short* __restrict__ out,
                                                        int N, int C, int H, int W, int P) {
  const int gh = H / P, gw = W / P, K = C * P * P, kv = K >> 3;
  const int64_t nvec = (int64_t)N * gh * gw * kv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int k0 = (int)(i % kv) * 8;
    const int64_t row = i / kv;
    const int px = (int)(row % gw), py = (int)((row / gw) % gh), n = (int)(row / ((int64_t)gw * gh));
    const int c = k0 / (P * P), r = k0 - c * P * P, ky = r / P, kx = r - ky * P;      // P % 8 == 0: 8 values share (c, ky)
    const float* src = img + (((size_t)n * C + c) * H + (py * P + ky)) * W + px * P + kx;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    *reinterpret_cast<u32x4*>(out + i * 8) = pack_h8(f);
  }
}

// ---------------------------------------------------------------- LayerNorm, rows of N = 8 * ACT halves
// LPR lanes per row (a power of two, for the shuffles), the first ACT of them active (N = 384: 48 of 64), 16 bytes per
// lane; y fp16 and / or y32 fp32 (the encoder's final norm hands fp32 tokens to the MIL head).
template <int LPR, int ACT>
__global__ __launch_bounds__(256) void layernorm_f16_kernel(const unsigned short* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             unsigned short* __restrict__ y, float* __restrict__ y32,
                                                             int64_t M, float eps) {
  constexpr int N = 8 * ACT;
  const int lane = threadIdx.x % LPR, rl = threadIdx.x / LPR, rls = 256 / LPR;
  const bool act = lane < ACT;
  const int col = (act ? lane : 0) * 8;
  float g[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { g[j] = gamma[col + j]; b[j] = beta[col + j]; }
  for (int64_t row = (int64_t)blockIdx.x * rls + rl; row < M; row += (int64_t)gridDim.x * rls) {
    float f[8];
    unpack_h8(*reinterpret_cast<const u32x4*>(x + row * N + col), f);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += act ? f[j] : 0.f;
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, LPR);
    const float mean = s * (1.f / N);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { f[j] -= mean; v += act ? f[j] * f[j] : 0.f; }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPR);
    const float rstd = rsqrtf(v * (1.f / N) + eps);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f[j] * rstd * g[j] + b[j];
    if (act && y) *reinterpret_cast<u32x4*>(y + row * N + col) = pack_h8(f);
    if (act && y32) {
      *reinterpret_cast<f32x4*>(y32 + row * N + col) = (f32x4){f[0], f[1], f[2], f[3]};
      *reinterpret_cast<f32x4*>(y32 + row * N + col + 4) = (f32x4){f[4], f[5], f[6], f[7]};
    }
  }
}

// ---------------------------------------------------------------- LayerNorm statistics only
// (mean, rstd) per row with the arithmetic of layernorm_f16_kernel, for a LayerNorm folded into the product that consumes it
// (isic_gemm_f16_ln): reads x once, writes 8 bytes per row.
template <int LPR, int ACT>
__global__ __launch_bounds__(256) void row_stats_f16_kernel(const unsigned short* __restrict__ x, float* __restrict__ stats,
                                                             int64_t M, float eps) {
  constexpr int N = 8 * ACT;
  const int lane = threadIdx.x % LPR, rl = threadIdx.x / LPR, rls = 256 / LPR;
  const bool act = lane < ACT;
  const int col = (act ? lane : 0) * 8;
  for (int64_t row = (int64_t)blockIdx.x * rls + rl; row < M; row += (int64_t)gridDim.x * rls) {
    float f[8];
    unpack_h8(*reinterpret_cast<const u32x4*>(x + row * N + col), f);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += act ? f[j] : 0.f;
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, LPR);
    const float mean = s * (1.f / N);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { f[j] -= mean; v += act ? f[j] * f[j] : 0.f; }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPR);
    if (lane == 0) {
      stats[row * 2] = mean;
      stats[row * 2 + 1] = rsqrtf(v * (1.f / N) + eps);
    }
  }
}

// ---------------------------------------------------------------- attention
// One block (8 waves) per (image, head): T <= 208 tokens, head width 64.  K and V row-major in LDS ([token][64] = 128-byte
// rows; 16-byte chunks XOR-swizzled by token & 7 for K, by (token >> 1) & 7 for V); Q is never staged: a wave reads the two
// 16-byte fragments of its 16 queries straight from global memory.  54 KB of LDS: two blocks per CU (register-limited), so
// one block's loads overlap the other's arithmetic.  A wave owns query tiles of 16: S^T = K.Q^T on
// v_mfma_f32_16x16x32_f16 (13 key tiles x 2 k-steps; a lane then holds, for ITS query l % 16, the scores of keys
// 16t + 4(l/16) + r), softmax over keys = in-lane over the 52 values + two xor shuffles; the probabilities are already
// in operand position of P.V up to a permutation of the contraction index (k-slot 8(l/16) + e <-> key
// 32u + 16(e/4) + 4(l/16) + e%4), which the V fragments follow.
// Round 3: O^T = V^T.P^T instead of O = P.V.  The V^T fragment (head dimension x keys) comes out of the row-major V with
// ds_read_b64_tr_b16 -- a 16-lane group addresses 4 keys x 16 head dimensions and each lane receives ONE dimension's four
// keys -- so V is staged with 16-byte writes like K (it was 32 ds_write_b16 per thread into a transposed copy); the head
// dimensions a group addresses are chosen so that a lane ends with EIGHT CONSECUTIVE dimensions of its own query
// (d = 32 (jd >> 1) + 8 (l / 16) + 4 (jd & 1) + r): the result is normalised with the lane's own 1 / sum and leaves in two
// 16-byte stores, no LDS transpose, no shuffles.
constexpr int AT_TMAX = 208, AT_KPAD = 224;                          // tokens padded to 13 x 16 (scores) / 7 x 32 (P.V)
constexpr int AT_WAVES = 8;
constexpr int AT_K = AT_TMAX * 128, AT_V = AT_KPAD * 128;
constexpr int AT_LDS = AT_K + AT_V;

__global__ __launch_bounds__(AT_WAVES * 64, 2) void attention_f16_kernel(const unsigned short* __restrict__ qkv,
                                                                          unsigned short* __restrict__ out, int T,
                                                                          int heads, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + AT_K;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const unsigned vs0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Vs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = blockIdx.x / heads, head = blockIdx.x - img * heads;
  const int D = heads * 64, ld = 3 * D;                               // qkv row: [q heads*64 | k heads*64 | v heads*64]
  const unsigned short* base = qkv + (size_t)img * T * ld + head * 64;
  const int fr = lane & 15, fg = lane >> 4;
  const int qtiles = (T + 15) >> 4;

  // ---- this wave's first query fragments (global -> registers), in flight while K and V are staged
  auto load_q = [&](int qt, f16x8 (&qf)[2]) {
    const int q = min(qt * 16 + fr, T - 1);                           // rows >= T: any valid row, never stored
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[ks] = *reinterpret_cast<const f16x8*>(base + (size_t)q * ld + (ks * 4 + fg) * 8);
  };
  f16x8 qf[2];
  if (wave < qtiles) load_q(wave, qf);

  // ---- stage K (swizzled rows of 128 B) and V^T; rows >= T are zero.  All of a thread's loads first, then the stores.
  constexpr int KV_PER = (AT_KPAD * 8 + AT_WAVES * 64 - 1) / (AT_WAVES * 64);      // 4 vectors of K and of V per thread
  u32x4 kr[KV_PER], vr[KV_PER];
#pragma unroll
  for (int u = 0; u < KV_PER; ++u) {
    const int idx = tid + u * AT_WAVES * 64, t = idx >> 3, ch = idx & 7;
    kr[u] = (u32x4){0u, 0u, 0u, 0u};
    vr[u] = kr[u];
    if (t < T) {
      kr[u] = *reinterpret_cast<const u32x4*>(base + (size_t)t * ld + D + ch * 8);
      vr[u] = *reinterpret_cast<const u32x4*>(base + (size_t)t * ld + 2 * D + ch * 8);
    }
  }
#pragma unroll
  for (int u = 0; u < KV_PER; ++u) {
    const int idx = tid + u * AT_WAVES * 64, t = idx >> 3, ch = idx & 7;
    if (t < AT_TMAX) *reinterpret_cast<u32x4*>(Ks + t * 128 + ((ch ^ (t & 7)) << 4)) = kr[u];
    if (t < AT_KPAD) *reinterpret_cast<u32x4*>(Vs + t * 128 + ((ch ^ ((t >> 1) & 7)) << 4)) = vr[u];
  }
  __syncthreads();

  // V^T fragment addresses: lane (fg, fq, fp) of a 16-lane group reads key 4fg + fq (+ 32u, + 16 for the upper half), the four
  // head dimensions 32 (jd >> 1) + 8 fp + 4 (jd & 1) + {0..3}: chunk (jd >> 1) * 4 + fp, swizzled by (key >> 1) & 7 (the
  // same for every u and both halves), byte (jd & 1) * 8 inside it.  16 keys x 4 x 8 bytes per instruction spread over half
  // of the banks evenly: the b64 rate.
  unsigned voff[4];
  {
    const int fq = fr >> 2, fp = fr & 3, key = 4 * fg + fq, hsw = (key >> 1) & 7;
#pragma unroll
    for (int jd = 0; jd < 4; ++jd)
      voff[jd] = vs0 + (unsigned)(key * 128 + ((((jd >> 1) * 4 + fp) ^ hsw) << 4) + (jd & 1) * 8);
  }
  for (int qt = wave; qt < qtiles; qt += AT_WAVES) {
    // ---- S^T tile row: keys x this tile's 16 queries
    f32x4 s[13];
#pragma unroll
    for (int kt = 0; kt < 13; ++kt) {
      const int krow = kt * 16 + fr;
      s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(Ks + krow * 128 + (((ks * 4 + fg) ^ (krow & 7)) << 4));
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], s[kt], 0, 0, 0);   // D[key 4fg+r][query fr]
      }
    }
    if (qt + AT_WAVES < qtiles) load_q(qt + AT_WAVES, qf);            // next tile's queries under the softmax
    // ---- softmax over the keys of query fr (base-2 exponent, scale folded in); keys >= T masked out
    // The softmax is what this kernel spends its issue slots on (52 values per lane and tile against 54 MFMAs): the maximum is
    // taken over the RAW scores (the scale is positive), scale and shift are one FMA, the exponential is the bare v_exp_f32
    // (results below 2^-126 flush to zero: they are rounded to fp16 next), and only a key tile that crosses T is masked.
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 13; ++kt) {
      if (kt * 16 + 16 > T) {                                         // wave-uniform: at most the last tiles
        asm volatile("" ::: "memory");                                // a real branch: if-converted, it costs 2 selects per value
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kt][r] = kt * 16 + fg * 4 + r < T ? s[kt][r] : -INFINITY;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float shift = -mx * scale_log2e;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 13; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], scale_log2e, shift));
        sum += s[kt][r];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;                                      // of query fr

    // ---- O^T = V^T.P^T: 7 key blocks of 32 (tiles 2u, 2u+1; tile 13 does not exist: zeros), 4 head-dimension tiles of 16
    f32x4 o[4];
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      f16x8 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[r] = (_Float16)s[2 * u][r];
        pf[4 + r] = (2 * u + 1 < 13) ? (_Float16)s[2 * u + 1 < 13 ? 2 * u + 1 : 12][r] : (_Float16)0.f;
      }
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {
        s16x8_t t;
        t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)(voff[jd] + (unsigned)(u * 32 * 128)));
        t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)(voff[jd] + (unsigned)(u * 32 * 128 + 16 * 128)));
        o[jd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, t), pf, o[jd], 0, 0, 0);   // D[d-slot 4fg+r][query fr]
      }
    }
    // ---- normalise with the lane's own 1 / sum; dimensions 32t' + 8fg + {0..7} of query fr: two 16-byte stores
    const int q = qt * 16 + fr;
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      u32x4 v;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 c = o[2 * tp + h] * inv;
        const f16x2 p0 = {(_Float16)c[0], (_Float16)c[1]}, p1 = {(_Float16)c[2], (_Float16)c[3]};
        v[2 * h] = __builtin_bit_cast(unsigned, p0);
        v[2 * h + 1] = __builtin_bit_cast(unsigned, p1);
      }
      if (q < T) *reinterpret_cast<u32x4*>(out + ((size_t)img * T + q) * D + head * 64 + 32 * tp + 8 * fg) = v;
    }
  }
}

}  // namespace

extern "C" {

int isic_vit_patchify_f16(const float* images_nchw, uint16_t* rows, int N, int C, int H, int W, int P, void* stream) {
  ISIC_CHECK_ARG(images_nchw && rows && N > 0 && C > 0 && H > 0 && W > 0 && P > 0);
  if (P % 8 != 0 || H % P != 0 || W % P != 0 || W % 4 != 0) return ISIC_ERR_UNSUPPORTED;
  const int64_t nvec = (int64_t)N * (H / P) * (W / P) * (C * P * P / 8);
  int64_t g = (nvec + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(patchify_kernel, dim3((int)g), dim3(256), 0, as_stream(stream), images_nchw, rows, N, C, H, W, P);
  return isic_launch_status();
}

int isic_layernorm_f16(const uint16_t* x, const float* gamma, const float* beta, uint16_t* y, float* y_f32, int64_t M,
                       int N, float eps, void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N > 0);
  if (M == 0) return ISIC_OK;
  ISIC_CHECK_ARG(x && gamma && beta && (y || y_f32));
  int64_t g;
#define LAUNCH_LN16(LPR, ACT)                                                                                    \
  g = (M + (256 / LPR) * 4 - 1) / ((256 / LPR) * 4);                                                             \
  if (g > 8192) g = 8192;                                                                                        \
  hipLaunchKernelGGL((layernorm_f16_kernel<LPR, ACT>), dim3((int)g), dim3(256), 0, as_stream(stream), x, gamma, beta, \
                     y, y_f32, M, eps)
  if (N == 128) { LAUNCH_LN16(16, 16); }
  else if (N == 256) { LAUNCH_LN16(32, 32); }
  else if (N == 384) { LAUNCH_LN16(64, 48); }
  else if (N == 512) { LAUNCH_LN16(64, 64); }
  else return ISIC_ERR_UNSUPPORTED;
#undef LAUNCH_LN16
  return isic_launch_status();
}

int isic_row_stats_f16(const uint16_t* x, float* stats, int64_t M, int N, float eps, void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N > 0);
  if (M == 0) return ISIC_OK;
  ISIC_CHECK_ARG(x && stats);
  int64_t g;
#define LAUNCH_RS16(LPR, ACT)                                                                                    \
  g = (M + (256 / LPR) * 4 - 1) / ((256 / LPR) * 4);                                                             \
  if (g > 8192) g = 8192;                                                                                        \
  hipLaunchKernelGGL((row_stats_f16_kernel<LPR, ACT>), dim3((int)g), dim3(256), 0, as_stream(stream), x, stats, M, eps)
  if (N == 128) { LAUNCH_RS16(16, 16); }
  else if (N == 256) { LAUNCH_RS16(32, 32); }
  else if (N == 384) { LAUNCH_RS16(64, 48); }
  else if (N == 512) { LAUNCH_RS16(64, 64); }
  else return ISIC_ERR_UNSUPPORTED;
#undef LAUNCH_RS16
  return isic_launch_status();
}

int isic_attention_f16(const uint16_t* qkv, uint16_t* out, int n_images, int tokens, int heads, int head_dim,
                       void* stream) {
  ISIC_CHECK_ARG(qkv && out && n_images > 0 && tokens > 0 && heads > 0);
  if (head_dim != 64 || tokens > AT_TMAX) return ISIC_ERR_UNSUPPORTED;
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (isic_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(attention_f16_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  const float scale_log2e = 0.125f * 1.4426950408889634f;            // 1 / sqrt(64), base-2 exponent
  hipLaunchKernelGGL(attention_f16_kernel, dim3(n_images * heads), dim3(AT_WAVES * 64), AT_LDS, as_stream(stream), qkv, out, tokens,
                     heads, scale_log2e);
  return isic_launch_status();
}

}  // extern "C"
