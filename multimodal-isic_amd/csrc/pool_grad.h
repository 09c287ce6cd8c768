// 3x3 / stride 2 / pad 1 max-pool gradient gathered on the fly, shared by the stem's BatchNorm backward kernels
// (encoder_ops.hip) and the stem weight gradient that consumes it without materialising anything (conv_stem.hip).
// NHWC bf16, 8 channels (16 bytes) per lane.  Every load uses a CLAMPED, always valid address and the validity is
// applied to the value afterwards: no load sits behind a branch, so all of a thread's loads are in flight together.
#pragma once
#include "common.h"

namespace isic_pool {

__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    v[i] = (unsigned)f32_to_bf16_bits(f[2 * i]) | ((unsigned)f32_to_bf16_bits(f[2 * i + 1]) << 16);
  return v;
}

struct PoolGeom { int H, W, C, Ho, Wo; };

// The four pooling windows (a+i, b+j), i, j in {0, 1}, that cover the 2x2 input pixels (2a+dy, 2b+dx): window (a, b)
// covers all four, (a, b+1) the right column, (a+1, b) the lower row, (a+1, b+1) the lower right pixel.
struct Windows {
  u32x4 g[4];      // pooled gradient, 8 channels
  u32x2 am[4];     // argmax codes kh*3+kw, one byte per channel
};
__device__ __forceinline__ void load_windows(Windows& w, const unsigned char* __restrict__ argmax,
                                             const unsigned short* __restrict__ gp, const PoolGeom& g_, int n, int a,
                                             int b, int cg) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jw = 0; jw < 2; ++jw) {
      const int ho = min(a + i, g_.Ho - 1), wo = min(b + jw, g_.Wo - 1);
      const int64_t o = (((int64_t)n * g_.Ho + ho) * g_.Wo + wo) * g_.C + cg * 8;
      w.am[i * 2 + jw] = *reinterpret_cast<const u32x2*>(argmax + o);
      w.g[i * 2 + jw] = *reinterpret_cast<const u32x4*>(gp + o);
    }
}
// gradient of the 2x2 pixels, each rounded to bf16 like a materialised max-pool backward would be; contributions are
// summed in the order of the materialising kernel (window rows, then columns)
__device__ __forceinline__ void windows_to_grad(const Windows& w, const PoolGeom& g_, int a, int b, float (&out)[4][8]) {
  float acc[4][8];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[p][j] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jw = 0; jw < 2; ++jw) {
      const bool live = (a + i < g_.Ho) && (b + jw < g_.Wo);
      float g[8];
      unpack8(w.g[i * 2 + jw], g);
      const u32x2 am = w.am[i * 2 + jw];
#pragma unroll
      for (int dy = i; dy < 2; ++dy)                    // window row i = 1 only reaches the lower pixels ...
#pragma unroll
        for (int dx = jw; dx < 2; ++dx) {               // ... window column 1 only the right ones
          const unsigned code = (unsigned)((dy + 1 - 2 * i) * 3 + (dx + 1 - 2 * jw));
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned bsel = (am[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
            if (live && bsel == code) acc[dy * 2 + dx][j] += g[j];
          }
        }
    }
#pragma unroll
  for (int p = 0; p < 4; ++p) unpack8(pack8(acc[p]), out[p]);
}

// the same for channels 4*HALF .. 4*HALF+3 only (lower register pressure where the caller holds a lot of other state)
template <int HALF>
__device__ __forceinline__ void windows_to_grad4(const Windows& w, const PoolGeom& g_, int a, int b, float (&out)[4][4]) {
  float acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[p][j] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jw = 0; jw < 2; ++jw) {
      const bool live = (a + i < g_.Ho) && (b + jw < g_.Wo);
      const unsigned lo = w.g[i * 2 + jw][2 * HALF], hi = w.g[i * 2 + jw][2 * HALF + 1];
      const float g[4] = {__uint_as_float(lo << 16), __uint_as_float(lo & 0xFFFF0000u), __uint_as_float(hi << 16),
                          __uint_as_float(hi & 0xFFFF0000u)};
      const unsigned am = w.am[i * 2 + jw][HALF];
#pragma unroll
      for (int dy = i; dy < 2; ++dy)
#pragma unroll
        for (int dx = jw; dx < 2; ++dx) {
          const unsigned code = (unsigned)((dy + 1 - 2 * i) * 3 + (dx + 1 - 2 * jw));
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (live && ((am >> (j * 8)) & 0xFFu) == code) acc[dy * 2 + dx][j] += g[j];
        }
    }
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) out[p][j] = bf16_bits_to_f32(f32_to_bf16_bits(acc[p][j]));
}

}  // namespace isic_pool
