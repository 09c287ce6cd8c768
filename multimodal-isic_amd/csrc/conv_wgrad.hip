// Convolution weight gradient on bf16 MFMA (gfx950).
//
//   dW[co, kh, kw, ci] = sum_{m=(n,ho,wo)} dY[m, co] * X[n, ho*s + kh - p, wo*s + kw - p, ci]
//
// GEMM view per tap (kh,kw): M = Cout, N = Cin, K = pixels.  Both operands are stored
// pixel-major with the channel contiguous (NHWC), i.e. the reduction index K is the SLOW
// dimension: the MFMA fragments (8 consecutive k per lane for one row/column) are produced by
// staging [64 pixels][channels] tiles in LDS untransposed (coalesced 16-byte global loads) and
// reading them back with ds_read_b64_tr_b16, gfx950's transposing LDS read (4 pixels x 16
// channels per 16-lane group, delivered channel-per-lane).  Rows are padded by 32 bytes so the 8
// pixel rows a 32-lane half touches land in disjoint banks.
//
// The K (pixel) range is split over blockIdx.y and partial results are accumulated with fp32
// atomics into the zero-initialised gradient (one 64-byte row segment per 16 lanes).
#include "common.h"

namespace {

struct WgradArgs {
  const unsigned short* x;
  const unsigned short* dy;
  float* dw;
  int N, Hin, Win, Cin, Hout, Wout, Cout, Kh, Kw, stride, pad;
  int M;               // N*Hout*Wout  (< 2^24)
  int ktiles;          // ceil(M/64)
  int ktiles_per_split;
  int co_tiles, ci_tiles;
  unsigned long long magic_hw, magic_w;  // floor(2^40/d)+1
};

__device__ __forceinline__ unsigned fastdiv40(unsigned n, unsigned long long magic) {
  return (unsigned)(((unsigned long long)n * magic) >> 40);
}

// TM x TN MFMA tiles per wave; block = 2x2 waves -> (32*TM) x (32*TN) outputs.
template <int TM, int TN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int BMc = 32 * TM, BNc = 32 * TN;        // channels per block (co / ci)
  constexpr int SA = BMc * 2 + 32, SB = BNc * 2 + 32;  // LDS row strides in bytes
  constexpr int BKP = 64;                              // pixels per K-tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[BKP * SA + BKP * SB];
  unsigned char* As = smem;
  unsigned char* Bs = smem + BKP * SA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // blockIdx.x = ((tap * ci_tiles) + ci_t) * co_tiles + co_t   (tiles sharing a pixel range run together)
  int bt = blockIdx.x;
  const int co_t = bt % a.co_tiles; bt /= a.co_tiles;
  const int ci_t = bt % a.ci_tiles;
  const int tap = bt / a.ci_tiles;
  const int kh = tap / a.Kw, kw = tap - kh * a.Kw;
  const int co0 = co_t * BMc, ci0 = ci_t * BNc;
  const int kt_begin = blockIdx.y * a.ktiles_per_split;
  int kt_end = kt_begin + a.ktiles_per_split;
  if (kt_end > a.ktiles) kt_end = a.ktiles;

  // staging: A rows have BMc*2/16 chunks, B rows BNc*2/16 chunks
  constexpr int ACH = BMc / 8, BCH = BNc / 8;
  constexpr int A_PER_T = BKP * ACH / 256, B_PER_T = BKP * BCH / 256;
  u32x4 ra[A_PER_T], rb[B_PER_T];
  const int HW = a.Hout * a.Wout;

  auto gload = [&](int kt) {
    const int mbase = kt * BKP;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int idx = tid + 256 * i, r = idx / ACH, ch = idx - r * ACH;
      const int m = mbase + r;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (m < a.M) v = *reinterpret_cast<const u32x4*>(a.dy + (size_t)m * a.Cout + co0 + ch * 8);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int idx = tid + 256 * i, r = idx / BCH, ch = idx - r * BCH;
      const int m = mbase + r;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (m < a.M) {
        const unsigned n = fastdiv40((unsigned)m, a.magic_hw);
        const unsigned rem = (unsigned)m - n * (unsigned)HW;
        const unsigned ho = fastdiv40(rem, a.magic_w);
        const unsigned wo = rem - ho * (unsigned)a.Wout;
        const int hi = (int)ho * a.stride + kh - a.pad, wi = (int)wo * a.stride + kw - a.pad;
        if (hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win)
          v = *reinterpret_cast<const u32x4*>(a.x + ((size_t)(n * a.Hin + hi) * a.Win + wi) * a.Cin + ci0 + ch * 8);
      }
      rb[i] = v;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int idx = tid + 256 * i, r = idx / ACH, ch = idx - r * ACH;
      *reinterpret_cast<u32x4*>(As + r * SA + ch * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int idx = tid + 256 * i, r = idx / BCH, ch = idx - r * BCH;
      *reinterpret_cast<u32x4*>(Bs + r * SB + ch * 16) = rb[i];
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fg = lane >> 4, fi = lane & 15, fq = fi >> 2, fp = fi & 3;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  if (kt_begin < kt_end) gload(kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (kt + 1 < kt_end) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int prow = ks * 32 + 8 * fg + fq;  // pixel row this lane addresses (first 4-row block)
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const unsigned char* p0 = As + prow * SA + ((wm * TM + i) * 16 + 4 * fp) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 4 * SA));
        s16x8_t t;
        t.lo = lo; t.hi = hi;
        af[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned char* p0 = Bs + prow * SB + ((wn * TN + j) * 16 + 4 * fp) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 4 * SB));
        s16x8_t t;
        t.lo = lo; t.hi = hi;
        bfr[j] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // accumulate: row = co, col = ci  ->  dw[co][kh][kw][ci]
  const size_t row_stride = (size_t)a.Kh * a.Kw * a.Cin;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wm * TM + i) * 16 + fg * 4 + r;
        const int ci = ci0 + (wn * TN + j) * 16 + fi;
        atomicAdd(a.dw + (size_t)co * row_stride + (size_t)tap * a.Cin + ci, acc[i][j][r]);
      }
}

}  // namespace

extern "C" {

int isic_conv2d_wgrad_bf16(const uint16_t* x, const uint16_t* dy, float* dw, int N, int Hin, int Win, int Cin,
                           int Hout, int Wout, int Cout, int Kh, int Kw, int stride, int pad, void* stream) {
  ISIC_CHECK_ARG(x && dy && dw);
  ISIC_CHECK_ARG(N > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Kh > 0 && Kw > 0 && stride > 0);
  if (Cin % 64 != 0 || Cout % 64 != 0) return ISIC_ERR_UNSUPPORTED;
  const int64_t M64 = (int64_t)N * Hout * Wout;
  if (M64 >= (1 << 24) || (int64_t)Hout * Wout >= (1 << 16)) return ISIC_ERR_UNSUPPORTED;  // fastdiv40 domain
  WgradArgs a;
  a.x = x; a.dy = dy; a.dw = dw;
  a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
  a.Kh = Kh; a.Kw = Kw; a.stride = stride; a.pad = pad;
  a.M = (int)M64; a.ktiles = ceil_div(a.M, 64);
  a.magic_hw = ((1ULL << 40) / (unsigned long long)(Hout * Wout)) + 1;
  a.magic_w = ((1ULL << 40) / (unsigned long long)Wout) + 1;
  const bool big = (Cin % 128 == 0) && (Cout % 128 == 0);
  const int tc = big ? 128 : 64;
  a.co_tiles = Cout / tc; a.ci_tiles = Cin / tc;
  const int tiles = a.co_tiles * a.ci_tiles * Kh * Kw;
  // enough K-splits for ~4 blocks per CU, each with at least 8 K-tiles
  int splits = ceil_div(1024, tiles);
  if (splits > ceil_div(a.ktiles, 8)) splits = ceil_div(a.ktiles, 8);
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  a.ktiles_per_split = ceil_div(a.ktiles, splits);
  splits = ceil_div(a.ktiles, a.ktiles_per_split);
  dim3 grid(tiles, splits);
  if (big) hipLaunchKernelGGL((conv_wgrad_kernel<4, 4>), grid, dim3(256), 0, as_stream(stream), a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<2, 2>), grid, dim3(256), 0, as_stream(stream), a);
  return isic_launch_status();
}

}  // extern "C"
