// Weight gradient of the 3x3 / STRIDE 2 / pad 1 layers (the first convolution of ResNet-18 layer2..4), all nine taps per
// block, the input staged ONCE per tile (round 4; conv_wgrad.hip's per-tap kernel keeps the other strided shapes):
//
//   dW[co][kh][kw][ci] = sum over output pixels (oy, ox) of  dY[oy][ox][co] * X[2 oy - 1 + kh][2 ox - 1 + kw][ci]
//
// The per-tap kernel stages a dY tile and the matching quarter of X once per tap: 9 x dY + 2.25 x X = 11 GB through the
// L2 -> LDS path for the 64 -> 128 layer at 4096 images, 1.5 ms at the ~6.7 TB/s that path delivers (335 TFLOP/s).  Here
//
//   * block = (64 input channels, 128 output channels, pixel range): 9 x 64 x 128 fp32 = 144 accumulator VGPRs per wave at
//     eight waves (wave = 16 input channels x 64 output channels), as in conv_wgrad_c128b.hip -- no staging waves, every
//     wave issues its eighth of a tile's 61 LDS-DMA groups between its MFMA groups;
//   * tile = 2 x 32 output pixels: 5 input rows x 65 input columns, staged DE-INTERLEAVED -- per patch row an odd-column
//     plane (40 positions: input columns -1, 1, 3, ...) and an even-column plane (32 positions) -- so that the 32 pixels of
//     a k-step are CONSECUTIVE positions of one plane for every tap column (kw = 0: odd plane from position 0, kw = 1: even
//     plane, kw = 2: odd plane from position 1) and the bank-conflict-free images of conv_wgrad_c128b.hip carry over;
//   * 46 KB of X + 16 KB of dY per tile and block for 72 MFMAs per wave: 3.6 GB staged per launch instead of 11;
//   * LDS images -- X: 128-byte position rows, 32-byte granule G of position q holds channel block
//     G ^ (((q >> 1) & 1) | ((q >> 3) & 1) << 1); dY: 256-byte pixel rows, granule G of tile pixel P holds channel block
//     G ^ ((P & 3) | ((P >> 3) & 1) << 2): the eight pixels b..b+3, b+8..b+11 a half-wave's transposing read touches land in
//     eight different 32-byte bank groups;
//   * two stages of 62,464 B; one barrier per tile; per-block partials + a fixed-order reduction: deterministic, no atomics;
//   * small images are packed two / four to a 32-column tile row (output widths <= 15 / <= 7) with at least one empty
//     column between them: its taps fall outside the image they belong to and read zeros.
//
// The reference has no convolution kernel of its own (un-vendored encoder, save_latent.py:42-60); ResNet-18 layer table:
// SURVEY.md 8d.
#include "common.h"

namespace {

constexpr int T_H = 2, T_W = 32;
constexpr int XROWS = 2 * T_H + 1;                  // 5 input rows per tile
constexpr int ODDP = 40, EVENP = 32;                // positions per plane (odd: input column 2 (q - 1) + 1, even: 2 q)
constexpr int XROWB = (ODDP + EVENP) * 128;         // 9,216 B per patch row
constexpr int XB = XROWS * XROWB;                   // 46,080 B
constexpr int YB = T_H * T_W * 256;                 // 16,384 B: [64 pixels][128 co]
constexpr int STG = XB + YB;                        // 62,464 B
constexpr int SCR = 2 * STG;
constexpr int LDS_ALL = SCR + 1024;                 // 125,952 B
constexpr int XGROUPS = XROWS * 9, YGROUPS = 16;    // 45 + 16 DMA groups (1 KB each) per tile
constexpr int NDMA = 8;                             // per wave (8 x 8 >= 61)
constexpr int SLICE_ELEMS = 128 * 9 * 64;           // one block's partial gradient

struct WS2Args {
  const unsigned short* x;      // [N][Hi][Wi][Cx]
  const unsigned short* dy;     // [N][Ho][Wo][Cy]
  float* partial;               // [pairs][blocks_per_pair][128][9][64], pair = ci_slice * (Cy / 128) + co_slice
  int N, Hi, Wi, Ho, Wo, tiles_y, tiles_x, total_tiles, tiles_per_block, blocks_per_pair;
  int Cx, Cy, co_slices;
  int pack, slot_shift, Wv;
};

__device__ __attribute__((aligned(256))) unsigned char g_ws2_zeros[2048];

__device__ __forceinline__ void glds16s(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(512) void wgrad_s2_kernel(WS2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = blockIdx.x / a.blocks_per_pair, bs = blockIdx.x - pair * a.blocks_per_pair;
  const int ci_slice = pair / a.co_slices, slice = pair - ci_slice * a.co_slices;      // slice: 128 output channels
  const int t_begin = bs * a.tiles_per_block;
  const int ntl = min(a.total_tiles - t_begin, a.tiles_per_block);       // >= 1 by construction of the grid

  // ---------------------------------------------------------------- staging (every wave: groups d = wave + 8 j)
  const int tiles_img = a.tiles_y * a.tiles_x;
  struct Tile { int n, y0, x0; };                                         // n: packed image group; (y0, x0): output pixel
  auto advance = [&](Tile& tl) {
    tl.x0 += T_W;
    if (tl.x0 >= a.Wv) {
      tl.x0 = 0; tl.y0 += T_H;
      if (tl.y0 >= a.Ho) { tl.y0 = 0; tl.n += 1; }
    }
  };
  Tile ahead;
  {
    const int n = t_begin / tiles_img, rem = t_begin - n * tiles_img;
    const int ty = rem / a.tiles_x;
    ahead.n = n; ahead.y0 = ty * T_H; ahead.x0 = (rem - ty * a.tiles_x) * T_W;
  }
  const int xl_px_ = lane >> 3, xl_slot = lane & 7;              // X: position in group, 16-byte slot of the 128-byte row
  const int yl_px_ = lane >> 4, yl_slot = lane & 15;             // dY: pixel in group, 16-byte slot of the 256-byte row
  // X position q = 8 gp + xl_px: key = ((q >> 1) & 1) | ((q >> 3) & 1) << 1 = ((xl_px >> 1) & 1) | (gp & 1) << 1
  const unsigned xsrc0 = (unsigned)(((((xl_slot >> 1) ^ ((xl_px_ >> 1) & 1)) << 1) | (xl_slot & 1)) << 4);
  const unsigned xsrc1 = (unsigned)(((((xl_slot >> 1) ^ (((xl_px_ >> 1) & 1) | 2)) << 1) | (xl_slot & 1)) << 4);
  // dY pixel P = 4 g2 + yl_px: key = (P & 3) | ((P >> 3) & 1) << 2 = yl_px | ((g2 >> 1) & 1) << 2
  const unsigned ysrc0 = (unsigned)(((((yl_slot >> 1) ^ yl_px_) << 1) | (yl_slot & 1)) << 4);
  const unsigned ysrc1 = (unsigned)(((((yl_slot >> 1) ^ (yl_px_ | 4)) << 1) | (yl_slot & 1)) << 4);
  const unsigned long long zeros = (unsigned long long)g_ws2_zeros;
  const int slot_mask = (1 << a.slot_shift) - 1;
  const int xpix = a.Cx * 2, ypix = a.Cy * 2;                    // bytes per pixel
  const unsigned long long xbase = (unsigned long long)a.x + (unsigned long long)ci_slice * 128;
  const unsigned long long ybase = (unsigned long long)a.dy + (unsigned long long)slice * 256;
  // one DMA group of tile `tl` into stage `stage` (j = 0..7; wave-uniform group number d = wave + 8 j)
  auto dma_one = [&](int j, const Tile& tl, int stage, bool live) {
    const int d = wave + 8 * j;
    // (the lane's share of an address depends on lane and wave only: hipcc would hoist it out of the tile loop for all
    //  groups at once and SPILL it -- a scratch reload waits vmcnt(0), i.e. for every DMA in flight; hide the invariance)
    int xl_px = xl_px_, yl_px = yl_px_;
    asm volatile("" : "+v"(xl_px), "+v"(yl_px));
    const unsigned sbase = lds0 + (unsigned)stage * STG;
    const int img0 = tl.n * a.pack;
    const int imgs_left = a.N - img0;
    unsigned long long src;
    unsigned dst;
    bool real = live;
    if (d < XGROUPS) {
      const int pr = d / 9, g = d - 9 * pr;
      const bool odd = g < 5;
      const int gp = odd ? g : g - 5;
      const int vj = tl.x0 + 8 * gp + xl_px - (odd ? 1 : 0);              // virtual output column of this position
      const int k = a.pack == 1 ? 0 : (vj >> a.slot_shift), rc = a.pack == 1 ? vj : (vj & slot_mask);
      const int xc = 2 * rc + (odd ? 1 : 0), xr = 2 * tl.y0 - 1 + pr;
      const bool ok = real && (unsigned)xr < (unsigned)a.Hi && vj >= 0 && (unsigned)xc < (unsigned)a.Wi &&
                      (unsigned)k < (unsigned)imgs_left && k < a.pack;
      const long long pix = ((long long)(img0 + k) * a.Hi + xr) * a.Wi + xc;
      src = (ok ? xbase + (unsigned long long)(pix * xpix) : zeros) + ((gp & 1) ? xsrc1 : xsrc0);
      dst = sbase + (unsigned)(pr * XROWB + g * 1024);
    } else if (d < XGROUPS + YGROUPS) {
      const int g2 = d - XGROUPS;
      const int r = g2 >> 3, c = 4 * (g2 & 7) + yl_px;
      const int vj = tl.x0 + c;
      const int k = a.pack == 1 ? 0 : (vj >> a.slot_shift), rc = a.pack == 1 ? vj : (vj & slot_mask);
      const bool ok = real && tl.y0 + r < a.Ho && rc < a.Wo && k < imgs_left && k < a.pack;
      const long long pix = ((long long)(img0 + k) * a.Ho + tl.y0 + r) * a.Wo + rc;
      src = (ok ? ybase + (unsigned long long)(pix * ypix) : zeros) + (((g2 >> 1) & 1) ? ysrc1 : ysrc0);
      dst = sbase + (unsigned)(XB + g2 * 1024);
    } else {
      real = false;
      src = zeros + (unsigned)(lane * 16);
      dst = 0;
    }
    glds16s(reinterpret_cast<const void*>(src), real ? dst : lds0 + SCR);
  };

  // ---------------------------------------------------------------- fragments: wave = (ci block c, co half hh)
  const int c = wave & 3, hh = wave >> 2;
  const int fg = lane >> 4, fi = lane & 15, fq = fi >> 2, fp = fi & 3;
  // X fragment of tap column kw, half h of the k-step: tile column p = 8 fg + fq + 4 h -> plane position p (+1 for kw = 2)
  unsigned xaddr[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = 8 * fg + fq + 4 * h + (kw == 2 ? 1 : 0);
      const int key = ((q >> 1) & 1) | (((q >> 3) & 1) << 1);
      xaddr[kw][h] = (unsigned)((kw == 1 ? ODDP * 128 : 0) + q * 128 + ((c ^ key) << 5) + fp * 8);
    }
  // dY, 16-channel block c2 of the wave's 64: tile pixel 32 s + 8 fg + fq (+ 4): key = fq | (fg & 1) << 2
  unsigned yaddr[4];
  {
    const int key = fq | ((fg & 1) << 2);
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) yaddr[c2] = (unsigned)(XB + (8 * fg + fq) * 256 + (((4 * hh + c2) ^ key) << 5) + fp * 8);
  }

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) acc[t][c2] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int j = 0; j < NDMA; ++j) dma_one(j, ahead, 0, true);
  advance(ahead);

  for (int kk = 0; kk < ntl; ++kk) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile kk landed (this wave's groups)
    __builtin_amdgcn_s_barrier();                       // ... every group; everyone is done with stage (kk + 1) & 1
    const unsigned st = lds0 + (unsigned)(kk & 1) * STG;
    const bool more = kk + 1 < ntl;
    const int nstage = (kk + 1) & 1;
    auto read_frag = [&](unsigned base_lo, unsigned base_hi, int off) -> bf16x8 {
      s16x8_t t;
      t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)(base_lo + (unsigned)off));
      t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)(base_hi + (unsigned)off));
      return __builtin_bit_cast(bf16x8, t);
    };
    bf16x8 xf[6][3];                            // X fragments of group (s, kh): patch row 2 s + kh, tap column kw
    bf16x8 yf[4];                               // dY fragments of the current k-step (one 32-pixel tile row), four co blocks
    auto read_x = [&](int grp) {
      const int pr = 2 * (grp / 3) + grp % 3;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xf[grp][kw] = read_frag(st + xaddr[kw][0], st + xaddr[kw][1], pr * XROWB);
    };
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) yf[c2] = read_frag(st + yaddr[c2], st + yaddr[c2] + 4 * 256, 0);
    read_x(0);
#pragma unroll
    for (int grp = 0; grp < 6; ++grp) {         // group = (k-step s, tap row kh): 3 tap columns x 4 co blocks = 12 MFMAs
      const int s = grp / 3, kh = grp - 3 * s;
      if (grp + 1 < 6) read_x(grp + 1);         // the next group's fragments are read under this group's MFMAs
      // the wave's eight DMA groups of the NEXT tile go out between the MFMA groups (two with each of the first two)
      if (grp < 2) { dma_one(2 * grp, ahead, nstage, more); dma_one(2 * grp + 1, ahead, nstage, more); }
      else dma_one(grp + 2, ahead, nstage, more);
#pragma unroll
      for (int c2 = 0; c2 < 4; ++c2) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          acc[kh * 3 + kw][c2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[grp][kw], yf[c2], acc[kh * 3 + kw][c2], 0, 0, 0);
        // the last group of k-step 0 hands each dY register set over to k-step 1 as soon as it is done with it
        if (grp == 2) yf[c2] = read_frag(st + yaddr[c2], st + yaddr[c2] + 4 * 256, 32 * 256);
      }
      __builtin_amdgcn_sched_barrier(0);        // a DMA's address arithmetic and the fragment reads stay in their own slot
    }
    advance(ahead);
  }

  // this block's partial: lane (fg, fi) holds D[ci = 16 c + 4 fg + r][co = 64 hh + 16 c2 + fi]
  float* part = a.partial + (size_t)blockIdx.x * SLICE_ELEMS;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2)
      *reinterpret_cast<f32x4*>(part + ((size_t)(hh * 64 + c2 * 16 + fi) * 9 + t) * 64 + c * 16 + fg * 4) = acc[t][c2];
}

// dw[128 co_slice + co][tap][64 ci_slice + ci] += sum over the pair's blocks (fixed order): thread (q, grp) sums blocks
// grp, grp+16, ... of four consecutive elements, the 16 group sums are combined through LDS in group order
__global__ __launch_bounds__(256) void wgrad_s2_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                               int blocks_per_pair, int co_slices, int Cin) {
  __shared__ f32x4 red[16][16];
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const size_t e4 = (size_t)blockIdx.x * 16 + q;                   // float4 index into [pairs][128][9][64]
  const int pair = (int)(e4 / (SLICE_ELEMS / 4));
  const size_t l4 = e4 - (size_t)pair * (SLICE_ELEMS / 4);         // ... inside the pair: (co * 9 + tap) * 16 + ci / 4
  const float* base = partial + (size_t)pair * blocks_per_pair * SLICE_ELEMS;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
  for (int b = grp; b < blocks_per_pair; b += 16) s0 += reinterpret_cast<const f32x4*>(base + (size_t)b * SLICE_ELEMS)[l4];
  red[grp][q] = s0;
  __syncthreads();
  if (threadIdx.x < 16) {
    f32x4 t = red[0][q];
#pragma unroll
    for (int g = 1; g < 16; ++g) t += red[g][q];
    const int ci_slice = pair / co_slices, co_slice = pair - ci_slice * co_slices;
    const int row = (int)(l4 >> 4), ci4 = (int)(l4 & 15);           // row = co * 9 + tap
    f32x4* out = reinterpret_cast<f32x4*>(dw + ((size_t)co_slice * 128 * 9 + row) * Cin + ci_slice * 64) + ci4;
    *out = *out + t;
  }
}

struct WS2Plan { int Ho, Wo, pack, slot_shift, Wv, tiles_y, tiles_x, total_tiles, tiles_per_block, blocks_per_pair, pairs; };

bool ws2_plan(int N, int Hi, int Wi, int Cin, int Cout, WS2Plan& p) {
  const int cus = isic_cu_count();
  if (Cin % 64 != 0 || Cout % 128 != 0 || N <= 0 || Hi <= 0 || Wi <= 0 || (Hi & 1) || (Wi & 1)) return false;
  p.Ho = Hi / 2; p.Wo = Wi / 2;
  p.pack = p.Wo <= 7 ? 4 : (p.Wo <= 15 ? 2 : 1);                  // >= 1 empty column between packed images
  p.slot_shift = p.pack == 4 ? 3 : (p.pack == 2 ? 4 : 5);
  p.Wv = p.pack == 1 ? p.Wo : T_W;
  p.tiles_y = ceil_div(p.Ho, T_H);
  p.tiles_x = ceil_div(p.Wv, T_W);
  const int64_t total = (int64_t)ceil_div(N, p.pack) * p.tiles_y * p.tiles_x;
  if (total > 0x7FFFFFFFLL || (int64_t)N * Hi * Wi * Cin > 0x7FFFFFFFFFLL || (int64_t)N * p.Ho * p.Wo * Cout > 0x7FFFFFFFFFLL)
    return false;
  p.total_tiles = (int)total;
  p.pairs = (Cin / 64) * (Cout / 128);
  const int per_pair = cus >= p.pairs ? cus / p.pairs : 1;
  p.tiles_per_block = (int)ceil_div64(total, per_pair);
  p.blocks_per_pair = (int)ceil_div64(total, p.tiles_per_block);
  return true;
}

}  // namespace

// bytes of workspace the strided all-taps kernel needs (0: shape not handled)
size_t isic_wgrad_s2_workspace_bytes(int N, int Hi, int Wi, int Cin, int Cout) {
  WS2Plan p;
  if (!ws2_plan(N, Hi, Wi, Cin, Cout, p)) return 0;
  return (size_t)p.pairs * p.blocks_per_pair * SLICE_ELEMS * sizeof(float);
}

// called by isic_conv2d_wgrad_bf16 for 3x3, stride 2, pad 1, even input sizes, Cin % 64 == 0, Cout % 128 == 0
int isic_wgrad_s2_launch(const uint16_t* x, const uint16_t* dy, float* dw, int N, int Hi, int Wi, int Cin, int Cout,
                         void* workspace, hipStream_t stream) {
  WS2Plan p;
  if (!ws2_plan(N, Hi, Wi, Cin, Cout, p)) return ISIC_ERR_UNSUPPORTED;
  WS2Args a;
  a.x = x; a.dy = dy; a.partial = reinterpret_cast<float*>(workspace);
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ho = p.Ho; a.Wo = p.Wo;
  a.tiles_y = p.tiles_y; a.tiles_x = p.tiles_x; a.total_tiles = p.total_tiles;
  a.tiles_per_block = p.tiles_per_block; a.blocks_per_pair = p.blocks_per_pair;
  a.Cx = Cin; a.Cy = Cout; a.co_slices = Cout / 128; a.pack = p.pack; a.slot_shift = p.slot_shift; a.Wv = p.Wv;
  static IsicPerDeviceOnce once;
  if (isic_once_per_device(once, [&] {
        return hipFuncSetAttribute((const void*)wgrad_s2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALL);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL(wgrad_s2_kernel, dim3(p.pairs * p.blocks_per_pair), dim3(512), LDS_ALL, stream, a);
  hipLaunchKernelGGL(wgrad_s2_reduce_kernel, dim3(p.pairs * (SLICE_ELEMS / 64)), dim3(256), 0, stream, a.partial, dw,
                     p.blocks_per_pair, a.co_slices, Cin);
  return ISIC_OK;
}
