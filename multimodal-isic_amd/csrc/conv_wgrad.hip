// Convolution weight gradient on bf16 MFMA (gfx950).
//
//   dW[co, kh, kw, ci] = sum_{m=(n,ho,wo)} dY[m, co] * X[n, ho*s + kh - p, wo*s + kw - p, ci]
//
// GEMM view per tap (kh,kw): M = Cout, N = Cin, K = pixels.  Both operands are stored
// pixel-major with the channel contiguous (NHWC), i.e. the reduction index K is the SLOW
// dimension: the MFMA fragments (8 consecutive k per lane for one row/column) are produced by
// staging [pixels][channels] tiles in LDS untransposed (coalesced 16-byte global loads) and
// reading them back with ds_read_b64_tr_b16, gfx950's transposing LDS read (4 pixels x 16
// channels per 16-lane group, delivered channel-per-lane).  Rows are padded by 32 bytes so the 8
// pixel rows a 32-lane half touches land in disjoint banks.
//
// Pixel -> source-address decoding is hoisted out of the hot loop into a per-call table
// (8 bytes per output pixel: element offset of tap (0,0) + packed (ho, wo)), built by a tiny
// kernel into the caller's workspace; a staged row then costs one 8-byte load, four compares and
// an add instead of two integer divisions.
//
// Two block shapes:
//   <KSPLIT=false> 128 co x 128 ci per block, 2x2 waves of 64x64, K-tile = 64 pixels;
//   <KSPLIT=true>   64 co x  64 ci per block (the 64-channel layers): all four waves own the whole
//                   64x64 tile (4x4 MFMA tiles each, same LDS-read : MFMA ratio as above) and split
//                   the 128-pixel K-tile four ways.
// The pixel range is additionally split over blockIdx.y; partial sums are accumulated with fp32
// atomics into the caller's gradient buffer.
#include "common.h"

namespace {

struct WgradArgs {
  const unsigned short* x;
  const unsigned short* dy;
  float* dw;
  const int2* tab;     // [M] : .x = ((n*Hin + ho*s - p)*Win + wo*s - p)*Cin, .y = ho << 16 | wo
  int N, Hin, Win, Cin, Hout, Wout, Cout, Kh, Kw, stride, pad;
  int M;               // N*Hout*Wout
  int ktiles;          // ceil(M / pixels-per-K-tile)
  int ktiles_per_split;
  int co_tiles, ci_tiles;
};

__global__ void wgrad_table_kernel(int2* __restrict__ tab, int M, int Hout, int Wout, int Hin, int Win, int Cin,
                                   int stride, int pad) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int hw = Hout * Wout;
  const int n = m / hw, r = m - n * hw;
  const int ho = r / Wout, wo = r - ho * Wout;
  int2 e;
  e.x = ((n * Hin + ho * stride - pad) * Win + wo * stride - pad) * Cin;
  e.y = (ho << 16) | wo;
  tab[m] = e;
}

template <bool KSPLIT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int BC = KSPLIT ? 64 : 128;              // channels per block (co and ci)
  constexpr int BKP = KSPLIT ? 128 : 64;             // pixels per K-tile
  constexpr int SROW = BC * 2 + 32;                  // LDS row stride in bytes
  constexpr int CH = BC / 8;                         // 16-byte chunks per row
  constexpr int PER_T = BKP * CH / 256;              // chunk loads per thread per operand (4)
  constexpr int RSTEP = 256 / CH;                    // rows between a thread's successive chunks
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BKP * SROW];
  unsigned char* As = smem;
  unsigned char* Bs = smem + BKP * SROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = KSPLIT ? 0 : (wave >> 1), wn = KSPLIT ? 0 : (wave & 1);
  // blockIdx.x = ((tap * ci_tiles) + ci_t) * co_tiles + co_t   (tiles sharing a pixel range run together)
  int bt = blockIdx.x;
  const int co_t = bt % a.co_tiles; bt /= a.co_tiles;
  const int ci_t = bt % a.ci_tiles;
  const int tap = bt / a.ci_tiles;
  const int kh = tap / a.Kw, kw = tap - kh * a.Kw;
  const int co0 = co_t * BC, ci0 = ci_t * BC;
  const int kt_begin = blockIdx.y * a.ktiles_per_split;
  int kt_end = kt_begin + a.ktiles_per_split;
  if (kt_end > a.ktiles) kt_end = a.ktiles;

  const int srow = tid / CH, sch = tid - srow * CH;          // this thread's first row / its chunk
  const int tapoff = (kh * a.Win + kw) * a.Cin + ci0 + sch * 8;
  const int hlo = a.pad - kh, wlo = a.pad - kw;              // valid iff hlo <= ho*s < Hin + hlo
  u32x4 ra[PER_T], rb[PER_T];
  int2 ent[PER_T];                       // table entries of the NEXT tile to be loaded (prefetched a tile early,
                                         // so the x loads never wait behind a dependent table load)
  auto tload = [&](int kt) {
    const int mbase = kt * BKP + srow;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int m = mbase + RSTEP * i;
      ent[i] = m < a.M ? a.tab[m] : make_int2(0, 0);
    }
  };
  auto gload = [&](int kt) {
    const int mbase = kt * BKP + srow;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int m = mbase + RSTEP * i;
      u32x4 va = {0u, 0u, 0u, 0u}, vb = {0u, 0u, 0u, 0u};
      if (m < a.M) {
        va = *reinterpret_cast<const u32x4*>(a.dy + (size_t)m * a.Cout + co0 + sch * 8);
        const int2 e = ent[i];
        const int hs = (e.y >> 16) * a.stride, ws = (e.y & 0xFFFF) * a.stride;
        if (hs >= hlo && hs < a.Hin + hlo && ws >= wlo && ws < a.Win + wlo)
          vb = *reinterpret_cast<const u32x4*>(a.x + (e.x + tapoff));
      }
      ra[i] = va; rb[i] = vb;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int r = srow + RSTEP * i;
      *reinterpret_cast<u32x4*>(As + r * SROW + sch * 16) = ra[i];
      *reinterpret_cast<u32x4*>(Bs + r * SROW + sch * 16) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fg = lane >> 4, fi = lane & 15, fq = fi >> 2, fp = fi & 3;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  constexpr int KSTEPS = KSPLIT ? 1 : 2;               // 32-pixel MFMA steps per wave per K-tile
  if (kt_begin < kt_end) { tload(kt_begin); gload(kt_begin); }
  if (kt_begin + 1 < kt_end) tload(kt_begin + 1);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (kt + 1 < kt_end) gload(kt + 1);
    if (kt + 2 < kt_end) tload(kt + 2);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int prow = (KSPLIT ? wave * 32 : ks * 32) + 8 * fg + fq;   // first pixel row this lane addresses
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned char* p0 = As + prow * SROW + ((wm * 4 + i) * 16 + 4 * fp) * 2;
        s16x8_t t;
        t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
        t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 4 * SROW));
        af[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned char* p0 = Bs + prow * SROW + ((wn * 4 + j) * 16 + 4 * fp) * 2;
        s16x8_t t;
        t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
        t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 4 * SROW));
        bfr[j] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // accumulate: row = co, col = ci  ->  dw[co][kh][kw][ci]
  const size_t row_stride = (size_t)a.Kh * a.Kw * a.Cin;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wm * 4 + i) * 16 + fg * 4 + r;
        const int ci = ci0 + (wn * 4 + j) * 16 + fi;
        atomicAdd(a.dw + (size_t)co * row_stride + (size_t)tap * a.Cin + ci, acc[i][j][r]);
      }
}

}  // namespace

extern "C" {

size_t isic_conv2d_wgrad_workspace_bytes(int N, int Hout, int Wout) {
  return (size_t)N * Hout * Wout * sizeof(int2) + 64;
}

int isic_conv2d_wgrad_bf16(const uint16_t* x, const uint16_t* dy, float* dw, int N, int Hin, int Win, int Cin,
                           int Hout, int Wout, int Cout, int Kh, int Kw, int stride, int pad, void* workspace,
                           size_t workspace_bytes, void* stream) {
  ISIC_CHECK_ARG(x && dy && dw && workspace);
  ISIC_CHECK_ARG(N > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Kh > 0 && Kw > 0 && stride > 0 && pad >= 0);
  if (Cin % 64 != 0 || Cout % 64 != 0) return ISIC_ERR_UNSUPPORTED;
  const int64_t M64 = (int64_t)N * Hout * Wout;
  if (M64 > 0x7FFFFFFFLL / 2 || (int64_t)N * Hin * Win * Cin > 0x7FFFFFFFLL || Hout >= 32768 || Wout >= 32768)
    return ISIC_ERR_UNSUPPORTED;   // 32-bit element offsets / 16-bit packed coordinates
  if (workspace_bytes < isic_conv2d_wgrad_workspace_bytes(N, Hout, Wout)) return ISIC_ERR_WORKSPACE;
  WgradArgs a;
  a.x = x; a.dy = dy; a.dw = dw; a.tab = reinterpret_cast<const int2*>(workspace);
  a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
  a.Kh = Kh; a.Kw = Kw; a.stride = stride; a.pad = pad;
  a.M = (int)M64;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(wgrad_table_kernel, dim3(ceil_div(a.M, 256)), dim3(256), 0, s, reinterpret_cast<int2*>(workspace),
                     a.M, Hout, Wout, Hin, Win, Cin, stride, pad);
  const bool big = (Cin % 128 == 0) && (Cout % 128 == 0);
  const int tc = big ? 128 : 64, bkp = big ? 64 : 128;
  a.ktiles = ceil_div(a.M, bkp);
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
  if (big) hipLaunchKernelGGL((conv_wgrad_kernel<false>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<true>), grid, dim3(256), 0, s, a);
  return isic_launch_status();
}

}  // extern "C"
