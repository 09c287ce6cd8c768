// Weight gradient of the 128 -> 128 channel 3x3 / stride 1 / pad 1 layers (ResNet-18 layer2) with all nine taps per
// block, on the plan of conv_wgrad_c64.hip (gfx950, bf16 MFMA, fp32 accumulate):
//
//   dW[co][kh][kw][ci] = sum over pixels p of  dY[p][co] * X[p + (kh-1, kw-1)][ci]
//
// The gradient of a 128 x 9 x 128 layer does not fit one block's registers, so the OUTPUT CHANNELS are split four ways:
// block (slice q, pixel range) accumulates dW[32q .. 32q+32][9][128] -- 36,864 fp32, 72 accumulator VGPRs per MFMA
// wave, as in the 64-channel kernel -- over a contiguous range of 4 x 32 pixel tiles.  Per tile four staging waves
// bring the (4+2) x (32+2) pixel halo patch of X (all 128 input channels, 256-byte rows) and the block's 32-channel
// slice of dY into LDS once, two stages deep; MFMA wave c owns input channels 16c .. 16c+16 for all taps and both
// 16-channel halves of the slice.  X is staged by each of the four slice blocks (L2 hits), dY only by its own:
// 121 staged bytes per MFMA against 250 for the one-tap-per-block kernel of conv_wgrad.hip.
// Per-block partials go to the workspace with plain stores and are summed in a fixed order: no atomics.
//
// Round 2: the same kernel serves every 3x3 / stride-1 layer with Cin % 128 == 0 and Cout % 32 == 0 (ResNet-18 layer3:
// 256 channels @ 14x14, layer4: 512 @ 7x7).  A block owns one (128-input-channel slice, 32-output-channel slice) PAIR --
// the gradient of a pair is exactly the 128 -> 128 problem on strided tensors -- and small images are PACKED into the
// 4 x 32 pixel tile: two 14-wide or four 7-wide images side by side in slots of 16 / 8 columns, the >= 1 empty columns
// between them staged as zeros (they are the images' padding).  Only the staging waves' address arithmetic knows; the
// MFMA waves see the same LDS images as before.  87.5 % of a tile's pixels are real at 28, 14 and 7 pixels alike.

#include "common.h"

namespace {

constexpr int T_H = 4, T_W = 32;                   // pixel tile
constexpr int XPITCH = 40;                          // LDS pitch of the X patch in pixels (10 DMA groups of 4; 9 are loaded)
constexpr int XROWS = T_H + 2;
constexpr int XB = XROWS * XPITCH * 256;            // 61,440 B
constexpr int YB = T_H * T_W * 64;                  // 8,192 B: [128 pixels][32 co]
constexpr int STG = XB + YB;                        // 69,632 B
constexpr int SCR = 2 * STG;                        // 1 KB landing zone of the padding DMAs
constexpr int LDS_ALL = SCR + 1024;                 // 140,288 B
constexpr int XGROUPS = XROWS * 9, YGROUPS = 8;     // 54 + 8 DMA groups per tile
constexpr int NDMA128 = 16;                         // per staging wave (4 x 16 >= 62)
constexpr int SLICE_ELEMS = 32 * 9 * 128;           // one block's partial gradient

struct WC128Args {
  const unsigned short* x;      // [N][H][W][Cx]
  const unsigned short* dy;     // [N][H][W][Cy]
  float* partial;               // [pairs][blocks_per_slice][32][9][128], pair = ci_slice * (Cy / 32) + co_slice
  int N, H, W, tiles_y, tiles_x, total_tiles, tiles_per_block, blocks_per_slice;
  int Cx, Cy;                   // channels of x / dy (pixel strides)
  int co_slices;                // Cy / 32
  int pack, slot_shift;         // images per tile row (1, 2, 4) and log2 of their slot width (5, 4, 3)
  int Wv;                       // width of the virtual image a tile row walks: W (pack 1) or 32
  int pairs;                    // (Cx / 128) * (Cy / 32)
  int xcd_group;                // 1: blocks that stage the SAME X tiles (one pixel range, every co slice) share an XCD's L2
};

__device__ __attribute__((aligned(256))) unsigned char g_wc128_zeros[2048];

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {   // see conv_wgrad.hip
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ABL (test builds only, include/isic_hip_test.h): bit 0 = no MFMAs, bit 1 = no fragment reads, bit 2 = no DMA -- where a tile's time goes
template <int ABL>
__global__ __launch_bounds__(768) void wgrad_c128_kernel(WC128Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (pair, pixel range).  Hardware deals blocks round-robin over the 8 XCDs (b and b + 8 share an L2).  The X tiles
  // of a pixel range are staged by EVERY co-slice block of that range (Cout / 32 of them: 4 / 8 / 16), the chip streams
  // HBM -> LDS at only ~10 B/clk per CU but L2 -> LDS at 40-50 (tests/probes/probe_stage.hip), and at 63 KB per 2,304
  // MFMA cycles the kernel asks for 27: with the co-slice blocks of a range on ONE XCD, walking their tiles in step (every
  // block is resident: one per CU), X comes out of L2 for all but the first of them.  Logical order: range-major, pair fastest.
  int pair, bs;
  if (a.xcd_group) {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);     // bijective
    bs = __builtin_amdgcn_readfirstlane(L / a.pairs);      // (integer division runs on the vector unit: back to a scalar)
    pair = L - bs * a.pairs;
  } else {
    pair = blockIdx.x / a.blocks_per_slice;
    bs = blockIdx.x - pair * a.blocks_per_slice;
  }
  const int ci_slice = pair / a.co_slices, slice = pair - ci_slice * a.co_slices;      // slice: 32 output channels
  const int t_begin = bs * a.tiles_per_block;
  const int ntl = min(a.total_tiles - t_begin, a.tiles_per_block);       // >= 1 by construction of the grid

  if (wave12 >= 8) {
    // =================================================================== staging waves (one per SIMD)
    // DMA group d = p + 4j of a tile.  d < 54: X patch row d / 9, pixels 4 (d % 9) .. +4 (256 bytes each: lane ->
    // pixel lane / 16, 16-byte slot lane % 16); 54 <= d < 62: dY pixels 16 (d - 54) .. +16 of the tile (64-byte slice
    // rows: lane -> pixel lane / 4, slot lane % 4); else padding.
    // X: the 32-byte granule G of patch pixel P = row * 40 + col holds input-channel block G ^ key(P),
    //    key = (P & 3) | ((P >> 3) & 1) << 2  (P & 3 = col & 3, (P >> 3) & 1 = (row + (col >> 3)) & 1);
    // dY: granule G of tile pixel P holds channel block G ^ ((P >> 3) & 1).  Both make the transposing reads of the
    //    MFMA waves (8 pixel rows b..b+3, b+8..b+11 per half-wave) conflict-free for every tap shift.
    const int p = wave12 - 8;
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
    const int xl_px = lane >> 4, xl_slot = lane & 15;              // X: pixel in group, 16-byte slot
    const int yl_px = lane >> 2, yl_slot = lane & 3;               // dY: pixel in group, 16-byte slot of the 64-byte slice row
    // byte offset of this lane's SOURCE chunk inside its pixel, for the two values of the wave-uniform key bit
    const unsigned xsrc0 = (unsigned)(((((xl_slot >> 1) ^ xl_px) << 1) | (xl_slot & 1)) << 4);          // key = px & 3 (bit 2 clear)
    const unsigned xsrc1 = (unsigned)(((((xl_slot >> 1) ^ (xl_px | 4)) << 1) | (xl_slot & 1)) << 4);    // bit 2 set
    const unsigned ysrc = (unsigned)(((((yl_slot >> 1) ^ ((yl_px >> 3) & 1)) << 1) | (yl_slot & 1)) << 4);
    const unsigned long long zeros = (unsigned long long)g_wc128_zeros;
    // virtual column vc of tile-row group tl.n -> (image, real column): slot k = vc >> slot_shift holds image
    // tl.n * pack + k, its columns 0 .. W-1 at the start of the slot; everything else is padding (zeros).
    // The staging waves sit on the critical path (62 DMAs per tile over four waves), so everything that does not depend
    // on the tile is computed ONCE per DMA group: the lane's slot, its byte offset from the tile's base pixel (32 bit),
    // its column inside the image; a tile then costs a wave-uniform 64-bit base, one 64-bit add and three compares.
    const int slot_mask = (1 << a.slot_shift) - 1;
    const int xpix = a.Cx * 2, ypix = a.Cy * 2;                      // bytes per pixel
    const unsigned long long xbase = (unsigned long long)a.x + (unsigned long long)ci_slice * 256;
    const unsigned long long ybase = (unsigned long long)a.dy + (unsigned long long)slice * 64;
    int g_off[NDMA128], g_col[NDMA128], g_slot[NDMA128];
    unsigned g_dst[NDMA128], g_src[NDMA128];
#pragma unroll
    for (int j = 0; j < NDMA128; ++j) {
      const int d = p + 4 * j;                                       // wave-uniform
      if (d < XGROUPS) {
        const int pr = d / 9, g = d - 9 * pr;
        const int vc = -1 + 4 * g + xl_px;                           // relative to the tile's first column
        const int k = a.pack == 1 ? 0 : (vc >> a.slot_shift), rc = a.pack == 1 ? vc : (vc & slot_mask);     // vc = -1: k = -1
        g_slot[j] = k; g_col[j] = rc;
        g_off[j] = ((k * a.H + (pr - 1)) * a.W + rc) * xpix;
        g_src[j] = ((pr + (g >> 1)) & 1) ? xsrc1 : xsrc0;            // key bit ((row + (col >> 3)) & 1), col = 4g + ..
        g_dst[j] = (unsigned)((pr * XPITCH + 4 * g) * 256);
      } else if (d < XGROUPS + YGROUPS) {
        const int g2 = d - XGROUPS;
        const int r = g2 >> 1, c = 16 * (g2 & 1) + yl_px;            // tile row / column of this lane's pixel
        const int k = a.pack == 1 ? 0 : (c >> a.slot_shift), rc = a.pack == 1 ? c : (c & slot_mask);
        g_slot[j] = k; g_col[j] = rc;
        g_off[j] = ((k * a.H + r) * a.W + rc) * ypix;
        g_src[j] = ysrc;
        g_dst[j] = (unsigned)(XB + g2 * 1024);
      } else {
        g_slot[j] = 0; g_col[j] = 0; g_off[j] = 0; g_src[j] = 0; g_dst[j] = 0;
      }
    }
    auto issue_tile = [&](const Tile& tl, int stage, bool live) {
      const unsigned sbase = lds0 + (unsigned)stage * STG;
      const long long porg = ((long long)tl.n * a.pack * a.H + tl.y0) * a.W + tl.x0;     // first pixel of the tile (slot 0)
      const unsigned long long xt = xbase + (unsigned long long)(porg * xpix), yt = ybase + (unsigned long long)(porg * ypix);
      const int imgs_left = a.N - tl.n * a.pack;                     // slots k < imgs_left hold an image
#pragma unroll
      for (int j = 0; j < NDMA128; ++j) {
        const int d = p + 4 * j;                                     // wave-uniform
        unsigned long long src;
        unsigned dst;
        bool real = live;
        if (d < XGROUPS) {
          const int pr = d / 9;
          const bool row_ok = (unsigned)(tl.y0 - 1 + pr) < (unsigned)a.H;                // wave-uniform
          const bool ok = real && row_ok && (unsigned)(tl.x0 + g_col[j]) < (unsigned)a.W && (unsigned)g_slot[j] < (unsigned)imgs_left &&
                          g_slot[j] < a.pack;
          src = (ok ? xt + (long long)g_off[j] : zeros) + g_src[j];
          dst = sbase + g_dst[j];
        } else if (d < XGROUPS + YGROUPS) {
          const int r = (d - XGROUPS) >> 1;
          const bool row_ok = tl.y0 + r < a.H;
          const bool ok = real && row_ok && tl.x0 + g_col[j] < a.W && g_slot[j] < imgs_left;
          src = (ok ? yt + (long long)g_off[j] : zeros) + g_src[j];
          dst = sbase + g_dst[j];
        } else {
          real = false;
          src = zeros + (unsigned)(lane * 16);
          dst = 0;
        }
        if (!(ABL & 4)) glds16(reinterpret_cast<const void*>(src), real ? dst : lds0 + SCR);
      }
    };
    issue_tile(ahead, 0, true);
    advance(ahead);
    for (int kk = 0; kk < ntl; ++kk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile kk landed (this wave's groups)
      __builtin_amdgcn_s_barrier();                       // ... every group; the MFMA waves are done with stage (kk+1)&1
      issue_tile(ahead, (kk + 1) & 1, kk + 1 < ntl);
      advance(ahead);
    }
    return;
  }

  // ======================================================================= MFMA waves: wave c = input channels 16c..16c+16
  const int wave = wave12;
  const int fg = lane >> 4, fi = lane & 15, fq = fi >> 2, fp = fi & 3;
  // fragment addresses (bytes inside a stage).  X, tap column kw, half h, parity of the patch row:
  // patch pixel column px = 8fg + fq + kw + 4h; key = (px & 3) | ((row + (px >> 3)) & 1) << 2
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
  // dY slice, 16-channel half c2: tile pixel s*32 + 8fg + fq (+4): key = (P >> 3) & 1 = fg & 1
  unsigned yaddr[2];
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2) yaddr[c2] = (unsigned)(XB + (8 * fg + fq) * 64 + ((c2 ^ (fg & 1)) << 5) + fp * 8);

  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kk = 0; kk < ntl; ++kk) {
    __builtin_amdgcn_s_barrier();              // the staging waves saw tile kk land
    const unsigned st = lds0 + (unsigned)(kk & 1) * STG;
    unsigned xb[3][2][2], yb[2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int h = 0; h < 2; ++h) { xb[kw][h][0] = st + xaddr[kw][h][0]; xb[kw][h][1] = st + xaddr[kw][h][1]; }
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) yb[c2] = st + yaddr[c2];
    auto read_frag = [&](unsigned base_lo, unsigned base_hi, int off) -> bf16x8 {
      s16x8_t t;
      if (ABL & 2) { t.lo = (s16x4){(short)base_lo, (short)off, 1, 2}; t.hi = t.lo; return __builtin_bit_cast(bf16x8, t); }
      t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)(base_lo + (unsigned)off));
      t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)(base_hi + (unsigned)off));
      return __builtin_bit_cast(bf16x8, t);
    };
    bf16x8 yf[4][2];                            // dY fragments of the four 32-pixel rows (k-steps) x two co halves
    bf16x8 xf[XROWS][3];                        // X fragments of patch row pr, tap column kw
    auto read_row = [&](int pr) {
      if (pr < T_H) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) yf[pr][c2] = read_frag(yb[c2], yb[c2] + 4 * 64, pr * 32 * 64);
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xf[pr][kw] = read_frag(xb[kw][0][pr & 1], xb[kw][1][pr & 1], pr * XPITCH * 256);
    };
    read_row(0);
#pragma unroll
    for (int pr = 0; pr < XROWS; ++pr) {        // patch row pr serves k-step s = pr - kh of tap row kh
      if (pr + 1 < XROWS) read_row(pr + 1);     // one row ahead of the MFMAs
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int s = pr - kh;
        if (s < 0 || s >= T_H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
            if (ABL & 1) { acc[kh * 3 + kw][c2][0] += __builtin_bit_cast(float, (int)(short)xf[pr][kw][0] + (int)(short)yf[s][c2][1]); }
            else acc[kh * 3 + kw][c2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[pr][kw], yf[s][c2], acc[kh * 3 + kw][c2], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // this block's partial: lane (fg, fi) holds D[ci = 16c + 4fg + r][co = 32 slice + 16 c2 + fi]
  float* part = a.partial + ((size_t)pair * a.blocks_per_slice + bs) * SLICE_ELEMS;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
      *reinterpret_cast<f32x4*>(part + ((size_t)(c2 * 16 + fi) * 9 + t) * 128 + wave * 16 + fg * 4) = acc[t][c2];
}

// dw[32 co_slice + co][tap][128 ci_slice + ci] += sum over the pair's blocks (fixed order): thread (q, grp) sums blocks
// grp, grp+16, ... of four consecutive elements, the 16 group sums are combined through LDS in group order
__global__ __launch_bounds__(256) void wgrad_c128_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                 int blocks_per_slice, int co_slices, int Cin) {
  __shared__ f32x4 red[16][16];
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const size_t e4 = (size_t)blockIdx.x * 16 + q;                   // float4 index into [pairs][32][9][128]
  const int pair = (int)(e4 / (SLICE_ELEMS / 4));
  const size_t l4 = e4 - (size_t)pair * (SLICE_ELEMS / 4);         // ... inside the pair: (co * 9 + tap) * 32 + ci / 4
  const float* base = partial + (size_t)pair * blocks_per_slice * SLICE_ELEMS;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
  for (int b = grp; b < blocks_per_slice; b += 16) s0 += reinterpret_cast<const f32x4*>(base + (size_t)b * SLICE_ELEMS)[l4];
  red[grp][q] = s0;
  __syncthreads();
  if (threadIdx.x < 16) {
    f32x4 t = red[0][q];
#pragma unroll
    for (int g = 1; g < 16; ++g) t += red[g][q];
    const int ci_slice = pair / co_slices, co_slice = pair - ci_slice * co_slices;
    const int row = (int)(l4 >> 5), ci4 = (int)(l4 & 31);           // row = co * 9 + tap
    f32x4* out = reinterpret_cast<f32x4*>(dw + ((size_t)co_slice * 32 * 9 + row) * Cin + ci_slice * 128) + ci4;
    *out = *out + t;
  }
}

struct WC128Plan { int pack, slot_shift, Wv, tiles_y, tiles_x, total_tiles, tiles_per_block, blocks_per_pair, pairs; };

bool wc128_plan(int N, int H, int W, int Cin, int Cout, WC128Plan& p) {
  const int cus = isic_cu_count();
  if (Cin % 128 != 0 || Cout % 32 != 0 || N <= 0 || H <= 0 || W <= 0) return false;
  p.pack = W <= 7 ? 4 : (W <= 15 ? 2 : 1);                        // >= 1 empty column between packed images
  p.slot_shift = p.pack == 4 ? 3 : (p.pack == 2 ? 4 : 5);
  p.Wv = p.pack == 1 ? W : T_W;
  p.tiles_y = ceil_div(H, T_H);
  p.tiles_x = ceil_div(p.Wv, T_W);
  const int64_t total = (int64_t)ceil_div(N, p.pack) * p.tiles_y * p.tiles_x;
  if (total > 0x7FFFFFFFLL || (int64_t)N * H * W * (Cin > Cout ? Cin : Cout) > 0x7FFFFFFFFFLL) return false;
  p.total_tiles = (int)total;
  p.pairs = (Cin / 128) * (Cout / 32);
  const int per_pair = cus >= p.pairs ? cus / p.pairs : 1;
  p.tiles_per_block = (int)ceil_div64(total, per_pair);
  p.blocks_per_pair = (int)ceil_div64(total, p.tiles_per_block);
  return true;
}

}  // namespace

// bytes of workspace the all-taps kernel needs for N images of H x W with Cin -> Cout channels (0: shape not handled)
size_t isic_wgrad_c128_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  WC128Plan p;
  if (!wc128_plan(N, H, W, Cin, Cout, p)) return 0;
  return (size_t)p.pairs * p.blocks_per_pair * SLICE_ELEMS * sizeof(float);
}

// called by isic_conv2d_wgrad_bf16 for 3x3, stride 1, pad 1, Cin % 128 == 0, Cout % 32 == 0
int isic_wgrad_c128_launch(const uint16_t* x, const uint16_t* dy, float* dw, int N, int H, int W, int Cin, int Cout,
                           void* workspace, int xcd_group, hipStream_t stream) {
  WC128Plan p;
  if (!wc128_plan(N, H, W, Cin, Cout, p)) return ISIC_ERR_UNSUPPORTED;
  WC128Args a;
  a.x = x; a.dy = dy; a.partial = reinterpret_cast<float*>(workspace);
  a.N = N; a.H = H; a.W = W;
  a.tiles_y = p.tiles_y; a.tiles_x = p.tiles_x; a.total_tiles = p.total_tiles;
  a.tiles_per_block = p.tiles_per_block; a.blocks_per_slice = p.blocks_per_pair;
  a.pairs = p.pairs; a.xcd_group = xcd_group;
  a.Cx = Cin; a.Cy = Cout; a.co_slices = Cout / 32; a.pack = p.pack; a.slot_shift = p.slot_shift; a.Wv = p.Wv;
  const int abl = (xcd_group >> 1) & 7;
  a.xcd_group = xcd_group & 1;
  const void* fns[8] = {(const void*)wgrad_c128_kernel<0>, (const void*)wgrad_c128_kernel<1>, (const void*)wgrad_c128_kernel<2>,
                        (const void*)wgrad_c128_kernel<3>, (const void*)wgrad_c128_kernel<4>, (const void*)wgrad_c128_kernel<5>,
                        (const void*)wgrad_c128_kernel<6>, (const void*)wgrad_c128_kernel<7>};
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device
  if (isic_once_per_device(once, [&] {
        hipError_t e = hipSuccess;
        for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALL);
        return e;
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  void* kargs[] = {&a};
  if (hipLaunchKernel(fns[abl], dim3(p.pairs * p.blocks_per_pair), dim3(768), kargs, LDS_ALL, stream) != hipSuccess) return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL(wgrad_c128_reduce_kernel, dim3(p.pairs * (SLICE_ELEMS / 64)), dim3(256), 0, stream, a.partial, dw,
                     p.blocks_per_pair, a.co_slices, Cin);
  return ISIC_OK;
}
