// Weight gradient of the 3x3 / stride 1 / pad 1 layers with Cin % 128 == 0 and Cout % 64 == 0 (ResNet-18 layer2..4), all
// nine taps per block, SIXTY-FOUR output channels per block (round 4; conv_wgrad_c128.hip keeps the 32-channel block for
// Cout % 64 != 0):
//
//   dW[co][kh][kw][ci] = sum over pixels p of  dY[p][co] * X[p + (kh-1, kw-1)][ci]
//
// What the 32-channel kernel waits for (round 4, compiled-out parts at 2048 images of 28 x 28 x 128, ms per launch): whole
// kernel 0.587 -- without MFMAs 0.587 -- without fragment reads 0.594 -- without LDS-DMA 0.322 -- DMA only 0.539.  It is the
// staging stream and nothing else: 63.5 KB per 4 x 32-pixel tile and block, 3.6 GB per launch at 6.7 TB/s, because the
// 55 KB halo patch of X is fetched again by every 32-output-channel block of a pixel range (Cout / 32 = 4 / 8 / 16 times).
// Putting those blocks on one XCD so that they share its L2 changed nothing (0.594 / 0.609 / 0.543 against 0.587 / 0.629 /
// 0.562 ms for 128 / 256 / 512 channels).  So the block is made to USE a staged patch twice as often instead:
//
//   * block = (128 input channels, 64 output channels, pixel range): 9 x 128 x 64 fp32 = 144 accumulator VGPRs per wave at
//     eight waves -- the whole register file of the CU (512 threads x 256), so there are no staging waves: every wave
//     multiplies AND issues its ninth of the tile's 70 LDS-DMA groups between its MFMAs (addresses formed on the fly);
//   * per tile 55 KB of X + 16 KB of dY for 144 MFMAs per wave instead of 63.5 KB for 72: 1.8x fewer staged bytes per MAC;
//   * LDS images as before -- X: 256-byte pixel rows, 32-byte granules XOR-swizzled by (P & 3) | ((row + (col >> 3)) & 1) << 2;
//     dY: now 128-byte pixel rows, granule G of tile pixel P holds channel block G ^ (((P >> 1) & 1) | ((P >> 3) & 1) << 1):
//     the eight pixels b..b+3, b+8..b+11 a half-wave's transposing read touches land in eight different 32-byte bank groups;
//   * two stages of 77,824 B; one barrier per tile; per-block partials + a fixed-order reduction: deterministic, no atomics;
//   * small images packed two / four to a 32-column tile row exactly as in conv_wgrad_c128.hip.
#include "common.h"

namespace {

constexpr int T_H = 4, T_W = 32;
constexpr int XPITCH = 40;
constexpr int XROWS = T_H + 2;
constexpr int XB = XROWS * XPITCH * 256;            // 61,440 B
constexpr int YB = T_H * T_W * 128;                 // 16,384 B: [128 pixels][64 co]
constexpr int STG = XB + YB;                        // 77,824 B
constexpr int SCR = 2 * STG;
constexpr int LDS_ALL = SCR + 1024;                 // 156,672 B
constexpr int XGROUPS = XROWS * 9, YGROUPS = 16;    // 54 + 16 DMA groups per tile
constexpr int NDMA = 9;                             // per wave (8 x 9 >= 70)
constexpr int SLICE_ELEMS = 64 * 9 * 128;           // one block's partial gradient

struct WC128BArgs {
  const unsigned short* x;      // [N][H][W][Cx]
  const unsigned short* dy;     // [N][H][W][Cy]
  float* partial;               // [pairs][blocks_per_pair][64][9][128], pair = ci_slice * (Cy / 64) + co_slice
  int N, H, W, tiles_y, tiles_x, total_tiles, tiles_per_block, blocks_per_pair;
  int Cx, Cy, co_slices;
  int pack, slot_shift, Wv;
};

__device__ __attribute__((aligned(256))) unsigned char g_wc128b_zeros[2048];

__device__ __forceinline__ void glds16b(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ABL (test entry only): bit 0 = no MFMAs, bit 1 = no fragment reads, bit 2 = no LDS-DMA -- where a tile's time goes
template <int ABL>
__global__ __launch_bounds__(512) void wgrad_c128b_kernel(WC128BArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = blockIdx.x / a.blocks_per_pair, bs = blockIdx.x - pair * a.blocks_per_pair;
  const int ci_slice = pair / a.co_slices, slice = pair - ci_slice * a.co_slices;      // slice: 64 output channels
  const int t_begin = bs * a.tiles_per_block;
  const int ntl = min(a.total_tiles - t_begin, a.tiles_per_block);       // >= 1 by construction of the grid

  // ---------------------------------------------------------------- staging (every wave: groups d = wave + 8 j)
  const int tiles_img = a.tiles_y * a.tiles_x;
  struct Tile { int n, y0, x0; };
  auto advance = [&](Tile& tl) {
    tl.x0 += T_W;
    if (tl.x0 >= a.Wv) {
      tl.x0 = 0; tl.y0 += T_H;
      if (tl.y0 >= a.H) { tl.y0 = 0; tl.n += 1; }
    }
  };
  Tile ahead;
  {
    const int n = t_begin / tiles_img, rem = t_begin - n * tiles_img;
    const int ty = rem / a.tiles_x;
    ahead.n = n; ahead.y0 = ty * T_H; ahead.x0 = (rem - ty * a.tiles_x) * T_W;
  }
  const int xl_px_ = lane >> 4, xl_slot = lane & 15;             // X: pixel in group, 16-byte slot of the 256-byte row
  const int yl_px_ = lane >> 3, yl_slot = lane & 7;              // dY: pixel in group, 16-byte slot of the 128-byte row
  const int xl_px = xl_px_, yl_px = yl_px_;
  const unsigned xsrc0 = (unsigned)(((((xl_slot >> 1) ^ xl_px) << 1) | (xl_slot & 1)) << 4);          // key bit 2 clear
  const unsigned xsrc1 = (unsigned)(((((xl_slot >> 1) ^ (xl_px | 4)) << 1) | (xl_slot & 1)) << 4);    // bit 2 set
  // dY pixel P = 32 r + 8 (g2 & 3) + yl_px: key = ((P >> 1) & 1) | ((P >> 3) & 1) << 1 = ((yl_px >> 1) & 1) | (g2 & 1) << 1
  const unsigned ysrc0 = (unsigned)(((((yl_slot >> 1) ^ ((yl_px >> 1) & 1)) << 1) | (yl_slot & 1)) << 4);
  const unsigned ysrc1 = (unsigned)(((((yl_slot >> 1) ^ (((yl_px >> 1) & 1) | 2)) << 1) | (yl_slot & 1)) << 4);
  const unsigned long long zeros = (unsigned long long)g_wc128b_zeros;
  const int slot_mask = (1 << a.slot_shift) - 1;
  const int xpix = a.Cx * 2, ypix = a.Cy * 2;                    // bytes per pixel
  const unsigned long long xbase = (unsigned long long)a.x + (unsigned long long)ci_slice * 256;
  const unsigned long long ybase = (unsigned long long)a.dy + (unsigned long long)slice * 128;
  // one DMA group of tile `tl` into stage `stage` (j = 0..8; wave-uniform group number d = wave + 8 j)
  auto dma_one = [&](int j, const Tile& tl, int stage, bool live) {
    const int d = wave + 8 * j;
    // (the lane's share of an address depends on lane and wave only: hipcc would hoist it out of the tile loop for all nine
    //  groups at once and SPILL it -- a scratch reload waits vmcnt(0), i.e. for every DMA in flight; hide the invariance)
    int xl_px = xl_px_, yl_px = yl_px_;
    asm volatile("" : "+v"(xl_px), "+v"(yl_px));
    const unsigned sbase = lds0 + (unsigned)stage * STG;
    const long long porg = ((long long)tl.n * a.pack * a.H + tl.y0) * a.W + tl.x0;     // first pixel of the tile (slot 0)
    const int imgs_left = a.N - tl.n * a.pack;
    unsigned long long src;
    unsigned dst;
    bool real = live;
    if (d < XGROUPS) {
      const int pr = d / 9, g = d - 9 * pr;
      const int vc = -1 + 4 * g + xl_px;
      const int k = a.pack == 1 ? 0 : (vc >> a.slot_shift), rc = a.pack == 1 ? vc : (vc & slot_mask);
      const bool row_ok = (unsigned)(tl.y0 - 1 + pr) < (unsigned)a.H;
      const bool ok = real && row_ok && (unsigned)(tl.x0 + rc) < (unsigned)a.W && (unsigned)k < (unsigned)imgs_left && k < a.pack;
      const long long off = (long long)(((k * a.H + (pr - 1)) * a.W + rc)) * xpix;
      src = (ok ? xbase + (unsigned long long)(porg * xpix + off) : zeros) + (((pr + (g >> 1)) & 1) ? xsrc1 : xsrc0);
      dst = sbase + (unsigned)((pr * XPITCH + 4 * g) * 256);
    } else if (d < XGROUPS + YGROUPS) {
      const int g2 = d - XGROUPS;
      const int r = g2 >> 2, c = 8 * (g2 & 3) + yl_px;
      const int k = a.pack == 1 ? 0 : (c >> a.slot_shift), rc = a.pack == 1 ? c : (c & slot_mask);
      const bool ok = real && tl.y0 + r < a.H && tl.x0 + rc < a.W && k < imgs_left;
      const long long off = (long long)(((k * a.H + r) * a.W + rc)) * ypix;
      src = (ok ? ybase + (unsigned long long)(porg * ypix + off) : zeros) + ((g2 & 1) ? ysrc1 : ysrc0);
      dst = sbase + (unsigned)(XB + g2 * 1024);
    } else {
      real = false;
      src = zeros + (unsigned)(lane * 16);
      dst = 0;
    }
    if (!(ABL & 4)) glds16b(reinterpret_cast<const void*>(src), real ? dst : lds0 + SCR);
  };

  // ---------------------------------------------------------------- fragments: wave c = input channels 16c .. 16c+16
  const int fg = lane >> 4, fi = lane & 15, fq = fi >> 2, fp = fi & 3;
  unsigned xaddr[3][2][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int px = 8 * fg + fq + kw + 4 * h;
        const int key = (px & 3) | (((par + (px >> 3)) & 1) << 2);
        xaddr[kw][h][par] = (unsigned)(px * 256 + ((wave ^ key) << 5) + fp * 8);
      }
  // dY, 16-channel block c2 of the 64: tile pixel 32 s + 8 fg + fq (+ 4): key = ((fq >> 1) & 1) | (fg & 1) << 1
  unsigned yaddr[4];
  {
    const int key = ((fq >> 1) & 1) | ((fg & 1) << 1);
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) yaddr[c2] = (unsigned)(XB + (8 * fg + fq) * 128 + ((c2 ^ key) << 5) + fp * 8);
  }

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
      if (ABL & 2) { t.lo = (s16x4){(short)base_lo, (short)off, 1, 2}; t.hi = t.lo; return __builtin_bit_cast(bf16x8, t); }
      t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)(base_lo + (unsigned)off));
      t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)(base_hi + (unsigned)off));
      return __builtin_bit_cast(bf16x8, t);
    };
    bf16x8 xf[XROWS][3];                        // X fragments of patch row pr, tap column kw (three rows live at a time)
    bf16x8 yf[4];                               // dY fragments of k-step s (one 32-pixel tile row), four co blocks
    auto read_xrow = [&](int pr) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        xf[pr][kw] = read_frag(st + xaddr[kw][0][pr & 1], st + xaddr[kw][1][pr & 1], pr * XPITCH * 256);
    };
    read_xrow(0);
    read_xrow(1);
#pragma unroll
    for (int s = 0; s < T_H; ++s) {             // k-step s = tile row s of dY against patch rows s, s+1, s+2 (tap rows 0, 1, 2)
#pragma unroll
      for (int c2 = 0; c2 < 4; ++c2) yf[c2] = read_frag(st + yaddr[c2], st + yaddr[c2] + 4 * 128, s * 32 * 128);
      read_xrow(s + 2);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        // the tile's nine DMA groups of the NEXT tile go out between the MFMA groups: 12 slots per tile, 9 used
        const int slot = s * 3 + kh;
        // (two per group in groups 0-4, or all nine in front of the first group: 0 .. +3 % slower -- when they go out is not it)
        if (slot < NDMA) dma_one(slot, ahead, nstage, more);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int c2 = 0; c2 < 4; ++c2)
            if (ABL & 1) acc[kh * 3 + kw][c2][0] += __builtin_bit_cast(float, (int)(short)xf[s + kh][kw][0] + (int)(short)yf[c2][1]);
            else acc[kh * 3 + kw][c2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[s + kh][kw], yf[c2], acc[kh * 3 + kw][c2], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);      // a DMA's address arithmetic and the fragment reads stay in their own slot
      }
    }
    advance(ahead);
  }

  // this block's partial: lane (fg, fi) holds D[ci = 16c + 4fg + r][co = 64 slice + 16 c2 + fi]
  float* part = a.partial + (size_t)blockIdx.x * SLICE_ELEMS;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2)
      *reinterpret_cast<f32x4*>(part + ((size_t)(c2 * 16 + fi) * 9 + t) * 128 + wave * 16 + fg * 4) = acc[t][c2];
}

// dw[64 co_slice + co][tap][128 ci_slice + ci] += sum over the pair's blocks (fixed order): thread (q, grp) sums blocks
// grp, grp+16, ... of four consecutive elements, the 16 group sums are combined through LDS in group order
__global__ __launch_bounds__(256) void wgrad_c128b_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                  int blocks_per_pair, int co_slices, int Cin) {
  __shared__ f32x4 red[16][16];
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const size_t e4 = (size_t)blockIdx.x * 16 + q;                   // float4 index into [pairs][64][9][128]
  const int pair = (int)(e4 / (SLICE_ELEMS / 4));
  const size_t l4 = e4 - (size_t)pair * (SLICE_ELEMS / 4);         // ... inside the pair: (co * 9 + tap) * 32 + ci / 4
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
    const int row = (int)(l4 >> 5), ci4 = (int)(l4 & 31);           // row = co * 9 + tap
    f32x4* out = reinterpret_cast<f32x4*>(dw + ((size_t)co_slice * 64 * 9 + row) * Cin + ci_slice * 128) + ci4;
    *out = *out + t;
  }
}

struct WC128BPlan { int pack, slot_shift, Wv, tiles_y, tiles_x, total_tiles, tiles_per_block, blocks_per_pair, pairs; };

bool wc128b_plan(int N, int H, int W, int Cin, int Cout, WC128BPlan& p) {
  const int cus = isic_cu_count();
  if (Cin % 128 != 0 || Cout % 64 != 0 || N <= 0 || H <= 0 || W <= 0) return false;
  p.pack = W <= 7 ? 4 : (W <= 15 ? 2 : 1);                        // >= 1 empty column between packed images
  p.slot_shift = p.pack == 4 ? 3 : (p.pack == 2 ? 4 : 5);
  p.Wv = p.pack == 1 ? W : T_W;
  p.tiles_y = ceil_div(H, T_H);
  p.tiles_x = ceil_div(p.Wv, T_W);
  const int64_t total = (int64_t)ceil_div(N, p.pack) * p.tiles_y * p.tiles_x;
  if (total > 0x7FFFFFFFLL || (int64_t)N * H * W * (Cin > Cout ? Cin : Cout) > 0x7FFFFFFFFFLL) return false;
  p.total_tiles = (int)total;
  p.pairs = (Cin / 128) * (Cout / 64);
  const int per_pair = cus >= p.pairs ? cus / p.pairs : 1;
  p.tiles_per_block = (int)ceil_div64(total, per_pair);
  p.blocks_per_pair = (int)ceil_div64(total, p.tiles_per_block);
  return true;
}

}  // namespace

// bytes of workspace the 64-output-channel all-taps kernel needs (0: shape not handled)
size_t isic_wgrad_c128b_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  WC128BPlan p;
  if (!wc128b_plan(N, H, W, Cin, Cout, p)) return 0;
  return (size_t)p.pairs * p.blocks_per_pair * SLICE_ELEMS * sizeof(float);
}

// called by isic_conv2d_wgrad_bf16 for 3x3, stride 1, pad 1, Cin % 128 == 0, Cout % 64 == 0
int isic_wgrad_c128b_launch(const uint16_t* x, const uint16_t* dy, float* dw, int N, int H, int W, int Cin, int Cout,
                            void* workspace, int ablation, hipStream_t stream) {
  WC128BPlan p;
  if (!wc128b_plan(N, H, W, Cin, Cout, p)) return ISIC_ERR_UNSUPPORTED;
  WC128BArgs a;
  a.x = x; a.dy = dy; a.partial = reinterpret_cast<float*>(workspace);
  a.N = N; a.H = H; a.W = W;
  a.tiles_y = p.tiles_y; a.tiles_x = p.tiles_x; a.total_tiles = p.total_tiles;
  a.tiles_per_block = p.tiles_per_block; a.blocks_per_pair = p.blocks_per_pair;
  a.Cx = Cin; a.Cy = Cout; a.co_slices = Cout / 64; a.pack = p.pack; a.slot_shift = p.slot_shift; a.Wv = p.Wv;
  const void* fns[8] = {(const void*)wgrad_c128b_kernel<0>, (const void*)wgrad_c128b_kernel<1>, (const void*)wgrad_c128b_kernel<2>,
                        (const void*)wgrad_c128b_kernel<3>, (const void*)wgrad_c128b_kernel<4>, (const void*)wgrad_c128b_kernel<5>,
                        (const void*)wgrad_c128b_kernel<6>, (const void*)wgrad_c128b_kernel<7>};
  static IsicPerDeviceOnce once;
  if (isic_once_per_device(once, [&] {
        hipError_t e = hipSuccess;
        for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALL);
        return e;
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  void* kargs[] = {&a};
  if (hipLaunchKernel(fns[ablation & 7], dim3(p.pairs * p.blocks_per_pair), dim3(512), kargs, LDS_ALL, stream) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL(wgrad_c128b_reduce_kernel, dim3(p.pairs * (SLICE_ELEMS / 64)), dim3(256), 0, stream, a.partial, dw,
                     p.blocks_per_pair, a.co_slices, Cin);
  return ISIC_OK;
}
