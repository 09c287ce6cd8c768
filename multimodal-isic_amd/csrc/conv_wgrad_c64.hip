// Weight gradient of the 64 -> 64 channel 3x3 / stride 1 / pad 1 layers (ResNet-18 layer1) with ALL NINE TAPS PER
// BLOCK and the whole 64 x 9 x 64 gradient resident in registers (gfx950, bf16 MFMA, fp32 accumulate).
//
//   dW[co][kh][kw][ci] = sum over pixels p of  dY[p][co] * X[p + (kh-1, kw-1)][ci]
//
// is a GEMM whose K dimension is the pixel index (1.6 M pixels at N = 512) and whose output is tiny.  The generic
// kernel (conv_wgrad.hip) gives every block one tap and re-reads dY and X from L2 nine times.  Here one persistent
// 768-thread block per CU walks a contiguous range of 4 x 32 pixel tiles; per tile four STAGING WAVES bring dY
// (16 KB) and the (4+2) x (32+2) pixel halo patch of X (26 KB used) into LDS ONCE by LDS-DMA, three stages deep, and
// MFMA wave (c, h) accumulates D[ci in 16c..16c+16][co in 32h..32h+32] for all nine taps: 18 MFMA tiles = 72
// accumulator VGPRs that never leave the register file until the block is done (two MFMA waves per SIMD: one wave
// alone cannot keep the matrix core busy between its own address arithmetic and LDS waits).  Every X fragment (patch row, tap column) is read from LDS once per tile
// and used for up to three tap rows; the operands are read with transposing LDS reads (ds_read_b64_tr_b16), since
// the K axis (pixels) is the row axis of the NHWC images in LDS.
// The partial gradients of the blocks go to the workspace with plain stores and are summed by a second kernel in
// a fixed order: the result does not depend on the scheduling (no atomics).

#include "common.h"

namespace {

constexpr int WT_H = 4, WT_W = 32;                  // pixel tile
constexpr int XP = 48;                              // pitch of the X patch in LDS (pixels; multiple of 16: swizzle key)
constexpr int X_ROWS = WT_H + 2;
constexpr int X_BYTES = X_ROWS * XP * 128;          // 36,864
constexpr int DY_BYTES = WT_H * WT_W * 128;         // 16,384
constexpr int STAGE = X_BYTES + DY_BYTES;           // 53,248
constexpr int NSTAGE = 3;
constexpr int SCRATCH = NSTAGE * STAGE;             // 1 KB landing zone of the padding DMAs
constexpr int LDS_TOTAL = SCRATCH + 1024;           // 160,768 B
constexpr int NDMA = 12;                            // DMA groups (8 pixels) per staging wave and tile: 6 + 4 + 2
constexpr int DW_ELEMS = 64 * 9 * 64;

struct WC64Args {
  const unsigned short* x;
  const unsigned short* dy;
  float* partial;       // [gridDim][64][9][64]
  int N, H, W, tiles_y, tiles_x, total_tiles, tiles_per_block;
};

__device__ __attribute__((aligned(256))) unsigned char g_wc64_zeros[2048];

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {   // see conv_wgrad.hip
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

#ifdef WC64_STAMPS   // tests/probes: per tile clock before the wait, after the barrier, after the MFMAs (wave 0, 256 blocks x 64 tiles)
__device__ unsigned long long g_wc64_stamps[256 * 64 * 4];
#define WC64_STAMP(kk, k) do { if (threadIdx.x == 512 && blockIdx.x < 256 && (kk) < 64) g_wc64_stamps[(blockIdx.x * 64 + (kk)) * 4 + (k)] = clock64(); } while (0)
#else
#define WC64_STAMP(kk, k) do { } while (0)
#endif

__global__ __launch_bounds__(768) void wgrad_c64_kernel(WC64Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t_begin = blockIdx.x * a.tiles_per_block;
  const int ntl = min(a.total_tiles - t_begin, a.tiles_per_block);       // >= 1 by construction of the grid

  if (wave12 >= 8) {
    // =================================================================== staging waves (one per SIMD)
    // An LDS-DMA instruction with per-lane addresses holds its wave at issue for the order of 100 cycles: issued by
    // the MFMA waves themselves, the twelve DMAs per SIMD and tile cost a quarter of the tile time.  These four waves
    // do nothing else, so the stall is theirs alone.
    // Wave p stages the 8-column block p of the X patch (6 rows) and of dY (4 rows), and rows 2p, 2p+1 of the X
    // patch's fifth block (columns 32..39; rows 6, 7 do not exist: padding): 12 DMA groups of 8 pixels per tile.
    // Lane -> pixel column offset lane/8, 16-byte slot lane%8.  The 32-byte granule G of the pixel in LDS column px
    // holds channel block G ^ key(px), key = bit1(px) | bit3(px) << 1: conflict-free transposing reads for every
    // tap shift (the 8 pixel rows a half-wave addresses are b..b+3 and b+8..b+11: bits 0, 1, 3 tell them apart).
    const int p = wave12 - 8;
    const int tiles_img = a.tiles_y * a.tiles_x;
    struct Tile { int n, y0, x0; };                    // wave-uniform tile coordinates, advanced incrementally
    auto advance = [&](Tile& tl) {
      tl.x0 += WT_W;
      if (tl.x0 >= a.W) {
        tl.x0 = 0; tl.y0 += WT_H;
        if (tl.y0 >= a.H) { tl.y0 = 0; tl.n += 1; }
      }
    };
    Tile ahead;
    {
      const int n = t_begin / tiles_img, rem = t_begin - n * tiles_img;
      const int ty = rem / a.tiles_x;
      ahead.n = n; ahead.y0 = ty * WT_H; ahead.x0 = (rem - ty * a.tiles_x) * WT_W;
    }
    const int r8 = lane >> 3, slot = lane & 7;
    const unsigned lane_e = (unsigned)(r8 * 128 + ((((slot >> 1) ^ ((r8 >> 1) & 1)) << 5) | ((slot & 1) << 4)));   // even blocks
    const unsigned lane_p = lane_e ^ (unsigned)((p & 1) << 6);                                                      // block p
    const long long W128 = (long long)a.W * 128;
    const unsigned long long zeros = (unsigned long long)g_wc64_zeros;
    auto issue_tile = [&](const Tile& tl, int stage, bool live) {
      const long long org = ((long long)tl.n * a.H + tl.y0) * a.W + tl.x0;               // pixel index of the tile origin
      const unsigned long long xrow = (unsigned long long)a.x + (unsigned long long)((org - a.W - 1 + 8 * p) * 128);   // patch (0, 8p)
      const unsigned long long yrow = (unsigned long long)a.dy + (unsigned long long)((org + 8 * p) * 128);            // tile (0, 8p)
      const unsigned long long erow = (unsigned long long)a.x + (unsigned long long)((org + (long long)(2 * p - 1) * a.W + 31) * 128);   // patch (2p, 32)
      const bool okx = (unsigned)(tl.x0 - 1 + 8 * p + r8) < (unsigned)a.W;
      const bool oky = (unsigned)(tl.x0 + 8 * p + r8) < (unsigned)a.W;
      const bool oke = (unsigned)(tl.x0 + 31 + r8) < (unsigned)a.W;
      const unsigned sbase = lds0 + (unsigned)stage * STAGE;
#pragma unroll
      for (int j = 0; j < NDMA; ++j) {
        unsigned long long rowaddr;
        bool okc, row_ok, real = live;
        unsigned lanev, dst;
        if (j < 6) {                                   // X patch row j, block p
          rowaddr = xrow + (unsigned long long)(j * W128);
          row_ok = (unsigned)(tl.y0 - 1 + j) < (unsigned)a.H; okc = okx; lanev = lane_p;
          dst = (unsigned)((j * XP + 8 * p) * 128);
        } else if (j < 10) {                           // dY row j - 6, block p
          rowaddr = yrow + (unsigned long long)((j - 6) * W128);
          row_ok = (unsigned)(tl.y0 + j - 6) < (unsigned)a.H; okc = oky; lanev = lane_p;
          dst = (unsigned)(X_BYTES + ((j - 6) * 4 + p) * 1024);
        } else {                                       // X patch row 2p + (j - 10), block 4
          const int pr = 2 * p + (j - 10);
          rowaddr = erow + (unsigned long long)((j - 10) * W128);
          row_ok = (unsigned)(tl.y0 - 1 + pr) < (unsigned)a.H; okc = oke; lanev = lane_e;
          real = real && pr < X_ROWS;
          dst = (unsigned)((pr * XP + 32) * 128);
        }
        const bool ok = okc && row_ok && real;
        // per-lane select of two wave-uniform addresses, half by half (out-of-image lanes read a zero page)
        const unsigned lo = ok ? (unsigned)rowaddr : (unsigned)zeros, hi = ok ? (unsigned)(rowaddr >> 32) : (unsigned)(zeros >> 32);
        const unsigned long long src = (((unsigned long long)hi << 32) | lo) + lanev;
        glds16(reinterpret_cast<const void*>(src), real ? sbase + dst : lds0 + SCRATCH);   // padding: zeros -> scratch
      }
    };
    issue_tile(ahead, 0, true);
    advance(ahead);
    issue_tile(ahead, 1, ntl > 1);
    advance(ahead);
    for (int kk = 0; kk < ntl; ++kk) {
      WC64_STAMP(kk, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");   // tile kk landed: only tile kk+1's groups in flight
      WC64_STAMP(kk, 3);
      __builtin_amdgcn_s_barrier();                   // the MFMA waves are done with stage (kk+2)%3 = (kk-1)%3
      WC64_STAMP(kk, 1);
      issue_tile(ahead, (kk + 2) % NSTAGE, kk + 2 < ntl);
      advance(ahead);
      WC64_STAMP(kk, 2);
    }
    return;
  }

  // ======================================================================= MFMA waves
  const int wave = wave12 & 3, half = wave12 >> 2;     // ci block / co half; waves (c, 0) and (c, 1) share a SIMD
  const int fg = lane >> 4, fi = lane & 15, fq = fi >> 2, fp = fi & 3;
  // ---- fragment addresses (bytes inside a stage).  A transposing read: lane (fg, fq, fp) addresses pixel row
  //      8fg + fq (+4 for the second half), 8 bytes at channel 4fp of a 16-channel block; lane fi then holds channel fi
  //      of pixels 8fg .. 8fg+3 (+4): the 8 consecutive K values of the 16x16x32 MFMA operands.
  //      X (ci block c), tap column kw: patch pixel column 8fg + fq + kw (+4)
  unsigned xaddr[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int px = 8 * fg + fq + kw + 4 * h;
      const int key = ((px >> 1) & 1) | (((px >> 3) & 1) << 1);
      xaddr[kw][h] = (unsigned)(px * 128 + ((wave ^ key) << 5) + fp * 8);
    }
  //      dY (co block 2h + c2): tile column 8fg + fq (+4): the key is the same for both halves
  unsigned yaddr[2];
  {
    const int key = ((fq >> 1) & 1) | ((fg & 1) << 1);
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) yaddr[c2] = (unsigned)(X_BYTES + (8 * fg + fq) * 128 + (((2 * half + c2) ^ key) << 5) + fp * 8);
  }

  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kk = 0; kk < ntl; ++kk) {
    __builtin_amdgcn_s_barrier();              // the staging waves saw tile kk land
    // LDS byte addresses of this tile's fragments
    const unsigned st = lds0 + (unsigned)(kk % NSTAGE) * STAGE;
    unsigned xb[3][2], yb[2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) { xb[kw][0] = st + xaddr[kw][0]; xb[kw][1] = st + xaddr[kw][1]; }
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) yb[c2] = st + yaddr[c2];
    auto read_frag = [&](unsigned base_lo, unsigned base_hi, int off) -> bf16x8 {
      s16x8_t t;
      t.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)(base_lo + (unsigned)off));
      t.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)(base_hi + (unsigned)off));
      return __builtin_bit_cast(bf16x8, t);
    };
    bf16x8 yf[4][2];                            // dY fragments of the four 32-pixel rows (k-steps) x this wave's two co blocks
    bf16x8 xf[X_ROWS][3];                       // X fragments of patch row pr, tap column kw
    auto read_row = [&](int pr) {
      if (pr < WT_H) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) yf[pr][c2] = read_frag(yb[c2], yb[c2] + 4 * 128, pr * 32 * 128);
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xf[pr][kw] = read_frag(xb[kw][0], xb[kw][1], pr * XP * 128);
    };
    read_row(0);
#pragma unroll
    for (int pr = 0; pr < X_ROWS; ++pr) {       // patch row pr serves k-step s = pr - kh of tap row kh
      if (pr + 1 < X_ROWS) read_row(pr + 1);    // one row ahead of the MFMAs
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int s = pr - kh;
        if (s < 0 || s >= WT_H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
            acc[kh * 3 + kw][c2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[pr][kw], yf[s][c2], acc[kh * 3 + kw][c2], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);        // keeps the prefetch distance: reads of row pr+1 stay above the MFMAs of row pr
    }
  }

  // ---- this block's partial gradient: lane (fg, fi) holds D[ci = 16c + 4fg + r][co = 32h + 16 c2 + fi]
  float* part = a.partial + (size_t)blockIdx.x * DW_ELEMS;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
      *reinterpret_cast<f32x4*>(part + ((size_t)(half * 32 + c2 * 16 + fi) * 9 + t) * 64 + wave * 16 + fg * 4) = acc[t][c2];
}

// dw[e] += sum over the blocks' partials, in a fixed order: thread (q, grp) of a block sums partials grp, grp+16, ...
// of four consecutive elements (16-byte loads); the 16 group sums are combined through LDS in group order.
__global__ __launch_bounds__(256) void wgrad_c64_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                int nblocks) {
  __shared__ f32x4 red[16][16];
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const size_t e4 = (size_t)blockIdx.x * 16 + q;                   // float4 index into the gradient
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  int b = grp;
  for (; b + 16 < nblocks; b += 32) {
    s0 += reinterpret_cast<const f32x4*>(partial + (size_t)b * DW_ELEMS)[e4];
    s1 += reinterpret_cast<const f32x4*>(partial + (size_t)(b + 16) * DW_ELEMS)[e4];
  }
  if (b < nblocks) s0 += reinterpret_cast<const f32x4*>(partial + (size_t)b * DW_ELEMS)[e4];
  red[grp][q] = s0 + s1;
  __syncthreads();
  if (threadIdx.x < 16) {
    f32x4 t = red[0][q];
#pragma unroll
    for (int g = 1; g < 16; ++g) t += red[g][q];
    f32x4* out = reinterpret_cast<f32x4*>(dw) + e4;
    *out = *out + t;
  }
}

int wc64_blocks(int N, int H, int W, int* tiles_per_block) {
  const int cus = isic_cu_count();
  const int64_t total = (int64_t)N * ceil_div(H, WT_H) * ceil_div(W, WT_W);
  const int tpb = (int)ceil_div64(total, cus);
  if (tiles_per_block) *tiles_per_block = tpb;
  return (int)ceil_div64(total, tpb);
}

}  // namespace

// bytes of workspace the 64 -> 64 kernel needs for N images of H x W (0: shape not handled)
size_t isic_wgrad_c64_workspace_bytes(int N, int H, int W) {
  if ((int64_t)N * ceil_div(H, WT_H) * ceil_div(W, WT_W) > 0x7FFFFFFFLL) return 0;
  return (size_t)wc64_blocks(N, H, W, nullptr) * DW_ELEMS * sizeof(float);
}

// called by isic_conv2d_wgrad_bf16 for Cin = Cout = 64, 3x3, stride 1, pad 1
int isic_wgrad_c64_launch(const uint16_t* x, const uint16_t* dy, float* dw, int N, int H, int W, void* workspace,
                          hipStream_t stream) {
  WC64Args a;
  a.x = x; a.dy = dy; a.partial = reinterpret_cast<float*>(workspace);
  a.N = N; a.H = H; a.W = W;
  a.tiles_y = ceil_div(H, WT_H); a.tiles_x = ceil_div(W, WT_W);
  a.total_tiles = N * a.tiles_y * a.tiles_x;
  const int grid = wc64_blocks(N, H, W, &a.tiles_per_block);
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (isic_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_TOTAL);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL(wgrad_c64_kernel, dim3(grid), dim3(768), LDS_TOTAL, stream, a);
  hipLaunchKernelGGL(wgrad_c64_reduce_kernel, dim3(DW_ELEMS / 64), dim3(256), 0, stream, a.partial, dw, grid);
  return ISIC_OK;
}
