// Launch arguments shared by the generic implicit-GEMM convolution kernels (conv_igemm.hip: one tile per block;
// conv_pgemm.hip: persistent blocks).  See conv_igemm.hip for the geometry they describe.
#pragma once
#include "common.h"

namespace isic_conv {

struct ConvArgs {
  const unsigned short* in;
  const unsigned short* w;
  unsigned short* out;
  const unsigned short* addend;
  double* stat_sum;
  double* stat_sumsq;
  int stat_slots;
  int N, Hin, Win, Cin, Hout, Wout, Cout, Kh, Kw, up, down_shift, pad;
  // output sub-grid of this launch: (ho, wo) = (oh0 + ostep*hs, ow0 + ostep*ws), hs < Hs, ws < Ws
  int Hs, Ws, oh0, ow0, ostep;
  // taps of this launch: kh = kh0 + kstep*i (i < nkh), kw = kw0 + kstep*j (j < nkw)
  int kh0, kw0, kstep, nkh, nkw;
  int M;        // N*Hs*Ws
  // a SECOND source joined into the same accumulators (round 4: the 1x1 / stride-2 downsample's data gradient inside the
  // 3x3 / stride-2 data gradient of a ResNet downsample block): K-tiles [Kmain, Ktiles) read `in2` (same geometry and
  // channel count as `in`) at the pixel of the class's tap (0, 0), with weights w2[Cout][Cin]; NULL / Kmain == Ktiles: none
  const unsigned short* in2;
  const unsigned short* w2;
  int Kmain;    // nkh*nkw*Cin/64
  int Ktiles;   // Kmain (+ Cin/64 for the class that owns the second source)
  int ctiles;   // Cin/64
  unsigned long long magic_hw, magic_w;   // floor(2^40/d)+1 for d = Hs*Ws and d = Ws (M < 2^24)
};

// the (up to four) output-parity classes of a strided data gradient run as ONE launch: blockIdx.z picks the class
struct ConvArgsN {
  ConvArgs c[4];
  int n;
};

__device__ __forceinline__ unsigned fastdiv40(unsigned n, unsigned long long magic) {
  return (unsigned)(((unsigned long long)n * magic) >> 40);
}
// bit t set iff lo <= base + t < hi, for t in [0, cnt)  (cnt <= 16)
__device__ __forceinline__ unsigned range_mask(int base, int lo, int hi, int cnt) {
  int t0 = lo - base, t1 = hi - base;
  t0 = t0 < 0 ? 0 : t0;
  t1 = t1 > cnt ? cnt : t1;
  return t1 > t0 ? (((1u << t1) - 1u) & ~((1u << t0) - 1u)) : 0u;
}

}  // namespace isic_conv
