// Convolution weight gradient on bf16 MFMA (gfx950).
//
//   dW[co, kh, kw, ci] = sum_{m=(n,ho,wo)} dY[m, co] * X[n, ho*s + kh - p, wo*s + kw - p, ci]
//
// GEMM view per tap (kh,kw): M = Cout, N = Cin, K = pixels.  Both operands are stored
// pixel-major with the channel contiguous (NHWC), i.e. the reduction index K is the SLOW
// dimension: the MFMA fragments (8 consecutive k per lane for one row/column) are produced by
// staging [pixels][channels] tiles in LDS untransposed (coalesced 16-byte global loads) and
// reading them back with ds_read_b64_tr_b16, gfx950's transposing LDS read (4 pixels x 16
// channels per 16-lane group, delivered channel-per-lane).  Staging is LDS-DMA (global_load_lds,
// two stages, no staging VGPRs); the 32-byte granules of a row are XOR-swizzled on the source side so
// the 8 pixel rows a 32-lane half touches land in disjoint banks.
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
// The pixel range is additionally split over blockIdx.y; a block's partial tile is transposed through
// LDS (and, with KSPLIT, summed over its four waves there) and written as whole 256-byte rows into ITS OWN slice of
// the workspace ([split][Cout][Kh][Kw][Cin]); wgrad_reduce_kernel then adds the slices to the caller's gradient in
// split order.  DETERMINISTIC (round 3): the fp32 row atomics this replaces added the splits in arrival order, so two
// identical steps gave different last bits in 6 of ResNet-18's 20 weight gradients.
#include "common.h"

namespace {

struct WgradArgs {
  const unsigned short* x;
  const unsigned short* dy;
  float* dw;           // split-K partials [splits][Cout][Kh][Kw][Cin] fp32 (workspace), overwritten
  size_t split_stride; // Cout*Kh*Kw*Cin
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

// One LDS-DMA instruction from inline asm: 64 lanes x 16 B from per-lane global addresses to the wave-uniform LDS
// byte address `lds_dst` (+ lane*16).  Issued from asm so that hipcc neither counts it in vmcnt nor fences the
// following ds_read_tr with a vmcnt(0) drain (it does for the builtin form in this kernel); completion is
// tracked by the loop's own s_waitcnt.  M0 (the DMA's LDS base) is saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// zero page: rows past the end of the pixel range and out-of-image taps are fetched from here
__device__ __attribute__((aligned(256))) unsigned char g_wgrad_zero_page[256];

template <bool KSPLIT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int BC = KSPLIT ? 64 : 128;              // channels per block (co and ci)
  constexpr int BKP = KSPLIT ? 128 : 64;             // pixels per K-tile
  constexpr int ROWB = BC * 2;                       // LDS row bytes (no padding: XOR swizzle on 32-byte granules)
  constexpr int OPB = BKP * ROWB;                    // bytes per operand tile (16 KB)
  constexpr int STAGE = 2 * OPB;
  constexpr int RPI = 1024 / ROWB;                   // rows per LDS-DMA wave-instruction (4 or 8)
  constexpr int PER_T = 4;                           // DMA instructions per wave per operand per K-tile
  constexpr int RSTEP = 4 * RPI;                     // rows between a lane's successive instructions (16 or 32)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = KSPLIT ? 0 : (wave >> 1), wn = KSPLIT ? 0 : (wave & 1);
  // blockIdx.x = ((tap * ci_tiles) + ci_t) * co_tiles + co_t   (tiles sharing a pixel range run together, spread over
  // the 8 XCDs; pinning all tiles of a pixel range to ONE XCD was measured slower: 256/512-channel layers have 36-144
  // tiles per range, more than an XCD runs at once, so they stop streaming in step -- l4 0.59 -> 0.88 ms)
  int bt = blockIdx.x;
  const int co_t = bt % a.co_tiles; bt /= a.co_tiles;
  const int ci_t = bt % a.ci_tiles;
  const int tap = bt / a.ci_tiles;
  const int kh = tap / a.Kw, kw = tap - kh * a.Kw;
  const int co0 = co_t * BC, ci0 = ci_t * BC;
  const int kt_begin = blockIdx.y * a.ktiles_per_split;
  int kt_end = kt_begin + a.ktiles_per_split;
  if (kt_end > a.ktiles) kt_end = a.ktiles;

  // LDS-DMA lane mapping: a wave-instruction writes 1 KB lane-linearly = RPI whole rows.  Lane l lands in row
  // (l / slots) at 16-byte slot (l % slots); the 32-byte granule q' = slot >> 1 of row r holds global granule
  // q' ^ swz(r).  A transposing read's 32-lane half touches pixel rows {b, b+1, b+2, b+3, b+8, .., b+11} (one
  // 32-byte granule each), so swz must separate exactly those 8 rows over the 256-byte bank row:
  //   256-byte rows: swz(r) = (r & 3) | ((r >> 3) & 1) << 2           (8 granules per row)
  //   128-byte rows: swz(r) = ((r >> 1) & 1) | ((r >> 3) & 1) << 1    (4 granules per row, 2 rows per bank row)
  // swz is the same for all of a lane's rows (they differ by multiples of 16 / 32).
  constexpr int SLOTS = ROWB / 16;
  const int lrow = wave * RPI + lane / SLOTS;                 // this lane's first row in the tile
  const int slot = lane % SLOTS;
  const int swz = KSPLIT ? (((lrow >> 1) & 1) | (((lrow >> 3) & 1) << 1)) : ((lrow & 3) | (((lrow >> 3) & 1) << 2));
  const int gchunk = 2 * ((slot >> 1) ^ swz) + (slot & 1);    // global 16-byte chunk of the row this lane fetches
  const int tapoff = (kh * a.Win + kw) * a.Cin + ci0 + gchunk * 8;
  const int hlo = a.pad - kh, wlo = a.pad - kw;              // valid iff hlo <= ho*s < Hin + hlo
  const unsigned char* zp = g_wgrad_zero_page + (lane & 7) * 16;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  // Table entries of the NEXT tile to be issued, prefetched one tile early.  These are ordinary register loads
  // living next to LDS-DMA traffic: if hipcc sees them it drains vmcnt(0) around them (and before the first
  // ds_read of the tile), which serialises DMA and MFMA.  They are therefore issued from inline asm (invisible
  // to the compiler's wait-count bookkeeping) and retired by the loop's own s_waitcnt vmcnt(0), whose asm
  // statement names them "+v" so that no use can be scheduled above it.
  unsigned long long ent[PER_T];
  auto tload = [&](int kt) {
    const int mbase = kt * BKP + lrow;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      int m = mbase + RSTEP * i;
      m = m < a.M ? m : a.M - 1;                              // always a valid address; validity is re-derived at issue
      const int2* p = a.tab + m;
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(ent[i]) : "v"(p) : "memory");
    }
  };
  auto twait = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ent[0]), "+v"(ent[1]), "+v"(ent[2]), "+v"(ent[3])::"memory");
  };
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // provably wave-uniform for the "s" operand
  const unsigned lds_base = (unsigned)(size_t)(lds_ptr)smem;  // LDS byte address of the staging area
  auto issue = [&](int kt, unsigned stage) {                   // stage = LDS byte address of the target stage
    const int mbase = kt * BKP + lrow;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int m = mbase + RSTEP * i;
      const int ex = (int)(unsigned)ent[i], ey = (int)(unsigned)(ent[i] >> 32);
      const bool inr = m < a.M;
      const int hs = (ey >> 16) * a.stride, ws = (ey & 0xFFFF) * a.stride;
      const bool okx = inr && hs >= hlo && hs < a.Hin + hlo && ws >= wlo && ws < a.Win + wlo;
      const void* sa = inr ? (const void*)(a.dy + (size_t)m * a.Cout + co0 + gchunk * 8) : (const void*)zp;
      const void* sb = okx ? (const void*)(a.x + (ex + tapoff)) : (const void*)zp;
      const unsigned dst = stage + (unsigned)((wave_u * RPI + RSTEP * i) * ROWB);
      glds16(sa, dst);
      glds16(sb, dst + OPB);
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
  // swizzle terms of the two 4-row blocks a lane addresses: rows (8*fg + fq) and (8*fg + fq + 4) (+ multiples of 32)
  const int r0_ = 8 * fg + fq, r1_ = r0_ + 4;
  const int sw0 = KSPLIT ? (((r0_ >> 1) & 1) | (((r0_ >> 3) & 1) << 1)) : ((r0_ & 3) | (((r0_ >> 3) & 1) << 2));
  const int sw1 = KSPLIT ? (((r1_ >> 1) & 1) | (((r1_ >> 3) & 1) << 1)) : ((r1_ & 3) | (((r1_ >> 3) & 1) << 2));
  if (kt_begin < kt_end) {
    tload(kt_begin);
    twait();
    issue(kt_begin, lds_base);
    if (kt_begin + 1 < kt_end) tload(kt_begin + 1);
  }
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    twait();                                           // tile kt (and the table entries of kt+1) landed
    __builtin_amdgcn_s_barrier();                      // ... for every wave; stage (kt+1)&1 is free again
    if (kt + 1 < kt_end) issue(kt + 1, lds_base + (unsigned)((((kt - kt_begin) + 1) & 1) * STAGE));
    if (kt + 2 < kt_end) tload(kt + 2);
    const unsigned char* As = smem + ((kt - kt_begin) & 1) * STAGE;
    const unsigned char* Bs = As + OPB;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int prow = (KSPLIT ? wave * 32 : ks * 32) + 8 * fg + fq;   // first pixel row this lane addresses
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cb = wm * 4 + i;
        s16x8_t t;
        t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(As + prow * ROWB + ((cb ^ sw0) << 5) + fp * 8));
        t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(As + (prow + 4) * ROWB + ((cb ^ sw1) << 5) + fp * 8));
        af[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cb = wn * 4 + j;
        s16x8_t t;
        t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Bs + prow * ROWB + ((cb ^ sw0) << 5) + fp * 8));
        t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Bs + (prow + 4) * ROWB + ((cb ^ sw1) << 5) + fp * 8));
        bfr[j] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: every wave parks its 64x64 fp32 partial tile in LDS ([co][ci], 16 KB per wave: exactly the
  // staging area), then whole 256-byte rows go out as ONE atomic wave-instruction each (the shape the memory-
  // side atomic units run at full rate).  With KSPLIT the four waves hold partials of the SAME tile: they are
  // summed here, so a block issues 64 row-atomics instead of 4 x 64 x 4 scattered ones.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // staging buffers are dead from here on
  float* Ct = reinterpret_cast<float*>(smem) + wave * (64 * 64);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ct[(i * 16 + fg * 4 + r) * 64 + j * 16 + fi] = acc[i][j][r];
  __syncthreads();
  const size_t row_stride = (size_t)a.Kh * a.Kw * a.Cin;
  const float* C0 = reinterpret_cast<const float*>(smem);
  float* part = a.dw + (size_t)blockIdx.y * a.split_stride;      // every (tile, split) block writes its whole tile
  if (KSPLIT) {
    for (int rr = 0; rr < 16; ++rr) {
      const int row = wave * 16 + rr;
      const float v = (C0[row * 64 + lane] + C0[4096 + row * 64 + lane]) + (C0[8192 + row * 64 + lane] + C0[12288 + row * 64 + lane]);
      part[(size_t)(co0 + row) * row_stride + (size_t)tap * a.Cin + ci0 + lane] = v;
    }
  } else {
    for (int row = 0; row < 64; ++row)
      part[(size_t)(co0 + wm * 64 + row) * row_stride + (size_t)tap * a.Cin + ci0 + wn * 64 + lane] = Ct[row * 64 + lane];
  }
}

// dw[e] += sum over the splits of partial[s][e], in split order.  G lanes per element share the splits (lane g takes
// s = g, g + G, ...) and meet in a fixed xor tree: the order never depends on timing.
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                            int splits, size_t elems) {
  const int g = threadIdx.x % G;
  const size_t e = (size_t)blockIdx.x * (256 / G) + threadIdx.x / G;
  float v = 0.f;
  if (e < elems)
    for (int s = g; s < splits; s += G) v += partial[(size_t)s * elems + e];
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
  if (e < elems && g == 0) dw[e] += v;
}

struct WgradPlan {
  bool big;
  int tiles, splits, ktiles, ktiles_per_split;
  size_t table_bytes, partial_bytes, elems;
};

WgradPlan wgrad_plan(int N, int Cin, int Hout, int Wout, int Cout, int Kh, int Kw) {
  WgradPlan p;
  const int M = N * Hout * Wout;
  p.big = (Cin % 128 == 0) && (Cout % 128 == 0);
  const int tc = p.big ? 128 : 64, bkp = p.big ? 64 : 128;
  p.ktiles = ceil_div(M, bkp);
  p.tiles = (Cout / tc) * (Cin / tc) * Kh * Kw;
  // two blocks per CU are resident (64 KB LDS each): ~2.5 rounds of 512 blocks, each with >= 8 K-tiles
  int splits = ceil_div(1280, p.tiles);
  if (splits > ceil_div(p.ktiles, 8)) splits = ceil_div(p.ktiles, 8);
  if (splits < 1) splits = 1;
  if (splits > 4096) splits = 4096;
  p.ktiles_per_split = ceil_div(p.ktiles, splits);
  p.splits = ceil_div(p.ktiles, p.ktiles_per_split);
  p.table_bytes = ((size_t)M * sizeof(int2) + 255) / 256 * 256;
  p.elems = (size_t)Cout * Kh * Kw * Cin;
  p.partial_bytes = (size_t)p.splits * p.elems * sizeof(float);
  return p;
}

}  // namespace

// all-taps-per-block kernel of the 64 -> 64 3x3 layers (conv_wgrad_c64.hip)
size_t isic_wgrad_c64_workspace_bytes(int N, int H, int W);
int isic_wgrad_c64_launch(const uint16_t* x, const uint16_t* dy, float* dw, int N, int H, int W, void* workspace,
                          hipStream_t stream);

// ... and of the 3x3 layers with Cin % 128 == 0, Cout % 32 == 0: 128, 256, 512 channels (conv_wgrad_c128.hip)
size_t isic_wgrad_c128_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int isic_wgrad_c128_launch(const uint16_t* x, const uint16_t* dy, float* dw, int N, int H, int W, int Cin, int Cout,
                           void* workspace, int xcd_group, hipStream_t stream);

// ... 64 output channels per block (conv_wgrad_c128b.hip): Cin % 128 == 0, Cout % 64 == 0
size_t isic_wgrad_c128b_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int isic_wgrad_c128b_launch(const uint16_t* x, const uint16_t* dy, float* dw, int N, int H, int W, int Cin, int Cout,
                            void* workspace, int ablation, hipStream_t stream);

// ... and of the 3x3 / stride 2 / pad 1 layers with Cin % 64 == 0, Cout % 128 == 0 (conv_wgrad_s2.hip)
size_t isic_wgrad_s2_workspace_bytes(int N, int Hi, int Wi, int Cin, int Cout);
int isic_wgrad_s2_launch(const uint16_t* x, const uint16_t* dy, float* dw, int N, int Hi, int Wi, int Cin, int Cout,
                         void* workspace, hipStream_t stream);

namespace {
// the all-taps kernels are always used for the shapes they cover (no environment switches, no global state)
inline bool wgrad_c128_enabled() { return true; }
constexpr int kWgradC128XcdGroup = 0;      // shipped block order of conv_wgrad_c128.hip (A/B: tools/halo_ab.py --wgrad)
inline bool wgrad_c64_enabled() { return true; }
}  // namespace

namespace {
int conv2d_wgrad_dispatch(const uint16_t* x, const uint16_t* dy, float* dw, int N, int Hin, int Win, int Cin,
                          int Hout, int Wout, int Cout, int Kh, int Kw, int stride, int pad, void* workspace,
                          size_t workspace_bytes, int variant, void* stream);
}

extern "C" {

size_t isic_conv2d_wgrad_workspace_bytes(int N, int Cin, int Hout, int Wout, int Cout, int Kh, int Kw) {
  if (N <= 0 || Cin <= 0 || Hout <= 0 || Wout <= 0 || Cout <= 0 || Kh <= 0 || Kw <= 0) return 0;
  const WgradPlan p = wgrad_plan(N, Cin, Hout, Wout, Cout, Kh, Kw);
  size_t need = p.table_bytes + p.partial_bytes + 256;
  if (Cin == 64 && Cout == 64 && Kh == 3 && Kw == 3) {   // stride 1 / pad 1: input size = output size
    const size_t c64 = isic_wgrad_c64_workspace_bytes(N, Hout, Wout);
    if (c64 > need) need = c64;
  }
  if (Cin % 128 == 0 && Cout % 32 == 0 && Kh == 3 && Kw == 3) {   // (a stride-2 layer of these widths asks for more than it uses)
    const size_t c128 = isic_wgrad_c128_workspace_bytes(N, Hout, Wout, Cin, Cout);
    if (c128 > need) need = c128;
    const size_t c128b = isic_wgrad_c128b_workspace_bytes(N, Hout, Wout, Cin, Cout);
    if (c128b > need) need = c128b;
  }
  if (Cin % 64 == 0 && Cout % 128 == 0 && Kh == 3 && Kw == 3) {   // the stride is not an argument: as if it were 2
    const size_t s2 = isic_wgrad_s2_workspace_bytes(N, 2 * Hout, 2 * Wout, Cin, Cout);
    if (s2 > need) need = s2;
  }
  return need;
}

int isic_conv2d_wgrad_bf16(const uint16_t* x, const uint16_t* dy, float* dw, int N, int Hin, int Win, int Cin,
                           int Hout, int Wout, int Cout, int Kh, int Kw, int stride, int pad, void* workspace,
                           size_t workspace_bytes, void* stream) {
  return conv2d_wgrad_dispatch(x, dy, dw, N, Hin, Win, Cin, Hout, Wout, Cout, Kh, Kw, stride, pad, workspace, workspace_bytes,
                               0, stream);
}

int isic_test_conv2d_wgrad_variant_bf16(const uint16_t* x, const uint16_t* dy, float* dw, int N, int Hin, int Win, int Cin,
                                        int Hout, int Wout, int Cout, int Kh, int Kw, int stride, int pad, void* workspace,
                                        size_t workspace_bytes, int variant, void* stream) {
  return conv2d_wgrad_dispatch(x, dy, dw, N, Hin, Win, Cin, Hout, Wout, Cout, Kh, Kw, stride, pad, workspace, workspace_bytes,
                               variant, stream);
}

}  // extern "C"

namespace {

// variant (include/isic_hip_test.h): 0 = shipped; bit 4 (16) = the 32-output-channel all-taps kernel where the 64-channel
// one ships, the per-tap kernel where the strided all-taps one ships; bit 0 = that kernel with the OTHER block order; bits 1-3 = its compiled-out parts (timing ablations)
int conv2d_wgrad_dispatch(const uint16_t* x, const uint16_t* dy, float* dw, int N, int Hin, int Win, int Cin,
                          int Hout, int Wout, int Cout, int Kh, int Kw, int stride, int pad, void* workspace,
                          size_t workspace_bytes, int variant, void* stream) {
  ISIC_CHECK_ARG(x && dy && dw && workspace);
  ISIC_CHECK_ARG(N > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Kh > 0 && Kw > 0 && stride > 0 && pad >= 0);
  ISIC_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0);
  if (Cin % 64 != 0 || Cout % 64 != 0) return ISIC_ERR_UNSUPPORTED;
  const int64_t M64 = (int64_t)N * Hout * Wout;
  if (M64 > 0x7FFFFFFFLL / 2 || (int64_t)N * Hin * Win * Cin > 0x7FFFFFFFLL || Hout >= 32768 || Wout >= 32768)
    return ISIC_ERR_UNSUPPORTED;   // 32-bit element offsets / 16-bit packed coordinates
  if (Cin == 64 && Cout == 64 && Kh == 3 && Kw == 3 && stride == 1 && pad == 1 && Hin == Hout && Win == Wout &&
      wgrad_c64_enabled()) {
    const size_t need = isic_wgrad_c64_workspace_bytes(N, Hin, Win);
    if (need != 0) {
      if (workspace_bytes < need) return ISIC_ERR_WORKSPACE;
      const int rc = isic_wgrad_c64_launch(x, dy, dw, N, Hin, Win, workspace, as_stream(stream));
      return rc != ISIC_OK ? rc : isic_launch_status();
    }
  }
  if (Cin % 128 == 0 && Cout % 64 == 0 && Kh == 3 && Kw == 3 && stride == 1 && pad == 1 && Hin == Hout && Win == Wout &&
      !(variant & 16)) {                                   // 64 output channels per block: 1.8x fewer staged bytes per MAC
    const size_t need = isic_wgrad_c128b_workspace_bytes(N, Hin, Win, Cin, Cout);
    if (need != 0) {
      if (workspace_bytes < need) return ISIC_ERR_WORKSPACE;
      const int rc = isic_wgrad_c128b_launch(x, dy, dw, N, Hin, Win, Cin, Cout, workspace, (variant >> 1) & 7, as_stream(stream));
      return rc != ISIC_OK ? rc : isic_launch_status();
    }
  }
  if (Cin % 128 == 0 && Cout % 32 == 0 && Kh == 3 && Kw == 3 && stride == 1 && pad == 1 && Hin == Hout && Win == Wout &&
      wgrad_c128_enabled()) {
    const size_t need = isic_wgrad_c128_workspace_bytes(N, Hin, Win, Cin, Cout);
    if (need != 0) {
      if (workspace_bytes < need) return ISIC_ERR_WORKSPACE;
      const int rc = isic_wgrad_c128_launch(x, dy, dw, N, Hin, Win, Cin, Cout, workspace, (kWgradC128XcdGroup ^ (variant & 1)) | (variant & 14),
                                            as_stream(stream));
      return rc != ISIC_OK ? rc : isic_launch_status();
    }
  }
  // the input staged once per tile instead of once per tap: 1.42 -> 0.71 ms for 64 -> 128 @ 56x56 and 0.76 -> 0.70 ms for
  // 128 -> 256 @ 28x28 at 4096 images; with 16 (64 ci, 128 co) pairs re-staging each other's operands (256 -> 512 @ 14x14)
  // it loses, 0.75 against 0.69 ms, and the per-tap kernel stays (variant bit 5 forces it on there: tools/halo_ab.py)
  if (Cin % 64 == 0 && Cout % 128 == 0 && Kh == 3 && Kw == 3 && stride == 2 && pad == 1 && Hin == 2 * Hout && Win == 2 * Wout &&
      !(variant & 16) && ((Cin / 64) * (Cout / 128) <= 4 || (variant & 32))) {
    const size_t need = isic_wgrad_s2_workspace_bytes(N, Hin, Win, Cin, Cout);
    if (need != 0) {
      if (workspace_bytes < need) return ISIC_ERR_WORKSPACE;
      const int rc = isic_wgrad_s2_launch(x, dy, dw, N, Hin, Win, Cin, Cout, workspace, as_stream(stream));
      return rc != ISIC_OK ? rc : isic_launch_status();
    }
  }
  const WgradPlan p = wgrad_plan(N, Cin, Hout, Wout, Cout, Kh, Kw);
  if (workspace_bytes < p.table_bytes + p.partial_bytes) return ISIC_ERR_WORKSPACE;
  WgradArgs a;
  float* partial = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + p.table_bytes);
  a.x = x; a.dy = dy; a.dw = partial; a.split_stride = p.elems; a.tab = reinterpret_cast<const int2*>(workspace);
  a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
  a.Kh = Kh; a.Kw = Kw; a.stride = stride; a.pad = pad;
  a.M = (int)M64;
  a.ktiles = p.ktiles; a.ktiles_per_split = p.ktiles_per_split;
  const int tc = p.big ? 128 : 64;
  a.co_tiles = Cout / tc; a.ci_tiles = Cin / tc;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(wgrad_table_kernel, dim3(ceil_div(a.M, 256)), dim3(256), 0, s, reinterpret_cast<int2*>(workspace),
                     a.M, Hout, Wout, Hin, Win, Cin, stride, pad);
  dim3 grid(p.tiles, p.splits);
  if (p.big) hipLaunchKernelGGL((conv_wgrad_kernel<false>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<true>), grid, dim3(256), 0, s, a);
  if (p.splits > 32)
    hipLaunchKernelGGL((wgrad_reduce_kernel<16>), dim3((unsigned)ceil_div64((int64_t)p.elems, 16)), dim3(256), 0, s, partial, dw,
                       p.splits, p.elems);
  else
    hipLaunchKernelGGL((wgrad_reduce_kernel<4>), dim3((unsigned)ceil_div64((int64_t)p.elems, 64)), dim3(256), 0, s, partial, dw,
                       p.splits, p.elems);
  return isic_launch_status();
}

}  // namespace

