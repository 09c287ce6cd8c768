// 3x3 / stride 1 / pad 1 convolution for the 64 -> 64 channel layers (ResNet-18 layer1, the largest
// activations of the network) with the INPUT PATCH RESIDENT IN LDS (gfx950, bf16 MFMA).
//
// The generic implicit GEMM re-fetches every input pixel once per tap: 9 x 128 B per output pixel
// from L2 for only 64 output channels -- 52 FLOP per staged byte, an L2-bandwidth-bound kernel.
// Here a block owns an 8 x 32 output tile of one image, stages the (8+2) x (32+2) pixel halo patch
// (43.5 KB) ONCE by LDS-DMA and reads all nine taps' A-fragments out of it with per-lane shifted
// ds_read_b128 addresses; only the 8 KB weight slice of the next tap streams in (two stages) while
// the current tap is multiplied: 3.1x fewer bytes staged per output pixel, 2 blocks per CU.
// Used for the forward pass and (flipped weights) the data gradient of those layers, with the same
// fused epilogues as conv_igemm.hip: residual-gradient addend, BatchNorm sum / sum-of-squares.
//
// Same operand conventions as conv_igemm.hip (NHWC bf16 activations, [64][3][3][64] bf16 weights,
// XOR-swizzled 128-byte LDS rows written lane-linearly by the DMA, D[co][pixel] accumulators).

#include "common.h"

namespace {

constexpr int TH = 8, TW = 32;                 // output tile (256 pixels = 4 waves x 64)
constexpr int PH = TH + 2, PW = TW + 2;        // halo patch 10 x 34 pixels
constexpr int PPIX = PH * PW;                  // 340
constexpr int PGROUPS = (PPIX + 7) / 8;        // 43 DMA groups of 8 pixels
constexpr int PATCH_BYTES = 44 * 1024;         // 4 waves x 11 groups x 1 KB
constexpr int WSTAGE = 64 * 128;               // one tap's weights: 64 co rows x 128 B
constexpr int CPAD = 64 + 8;
constexpr int WSTAGES = 4;                     // weight ring: three taps in flight behind the one being multiplied
constexpr int LDS_BYTES = PATCH_BYTES + WSTAGES * WSTAGE;   // 77,824 B -> 2 blocks per CU

struct C64Args {
  const unsigned short* in;
  const unsigned short* w;       // [64][3][3][64]
  unsigned short* out;
  const unsigned short* addend;
  double* stat_sum;
  double* stat_sumsq;
  int stat_slots;
  int N, H, W, tiles_y, tiles_x;
};

__device__ __attribute__((aligned(256))) unsigned char g_c64_zero_page[256];

__device__ __forceinline__ float bfb(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// 16-byte-per-lane LDS-DMA in inline asm (see conv_wgrad.hip): keeps the loads out of hipcc's vmcnt bookkeeping,
// the tap loop below counts them itself.  M0 (the DMA's LDS base) is saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

#ifdef C64_STAMPS   // tests/probes/probe_c64_stamps.hip: per-block phase timestamps
__device__ unsigned long long g_c64_stamps[8192 * 4];
#define C64_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_c64_stamps[blockIdx.x * 4 + (k)] = clock64(); } while (0)
#else
#define C64_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(256) void conv3x3_c64_kernel(C64Args a) {
  C64_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* patch = smem;
  unsigned char* wst = smem + PATCH_BYTES;
  const unsigned patch_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)patch;
  const unsigned wst_lds = patch_lds + PATCH_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  int bt = blockIdx.x;
  const int tx = bt % a.tiles_x; bt /= a.tiles_x;
  const int ty = bt % a.tiles_y;
  const int n = bt / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;

  // ---- stage the halo patch: DMA group g = 8 consecutive patch pixels (1 KB), lane -> (pixel 8g + l/8, slot l%8),
  //      slot p of pixel q holds global chunk p ^ (q & 7)
  const int r8 = lane >> 3, gch = (lane & 7) ^ r8;
  const unsigned char* zp = g_c64_zero_page + (lane & 7) * 16;
#pragma unroll
  for (int j = 0; j < 11; ++j) {
    const int g = wave + 4 * j;
    const int pp = g * 8 + r8;
    const int py = pp / PW, px = pp - py * PW;
    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
    const bool ok = pp < PPIX && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    const void* src = ok ? (const void*)(a.in + (((size_t)n * a.H + yy) * a.W + xx) * 64 + gch * 8) : (const void*)zp;
    glds16(src, patch_lds + g * 1024);
  }
  // weights of one tap: 64 rows x 128 B = 8 DMA groups, 2 per wave; row co = 8g + l/8, chunk (l%8) ^ (co&7) = gch
  const unsigned short* wsrc = a.w + (size_t)(wave * 8 + r8) * 576 + gch * 8;
  auto issue_w = [&](int tap) {
    const unsigned stage = wst_lds + (tap % WSTAGES) * WSTAGE;
    glds16(wsrc + tap * 64, stage + wave * 1024);
    glds16(wsrc + 32 * 576 + tap * 64, stage + (wave + 4) * 1024);
  };
  issue_w(0); issue_w(1); issue_w(2);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this lane's 4 output pixels (one per M-tile): tile row 2*wave + (i>>1), tile column (i&1)*16 + fr;
  // patch pixel of tap (0,0) = the same coordinates (the patch origin is the tile origin - 1)
  int pp0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pp0[i] = (2 * wave + (i >> 1)) * PW + (i & 1) * 16 + fr;

#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    // the patch and the weights of this tap have landed: only the (up to two) later taps' loads may be in flight
    if (tap <= 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (tap == 7) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // ... for every wave; and stage (tap-1)%4 is free again
    if (tap == 0) C64_STAMP(1);
    if (tap + 3 < 9) issue_w(tap + 3);
    const unsigned char* Bs = wst + (tap % WSTAGES) * WSTAGE;
    const int kh = tap / 3, kw = tap - kh * 3;
    const int poff = kh * PW + kw;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pp = pp0[i] + poff;
        af[i] = *reinterpret_cast<const bf16x8*>(patch + pp * 128 + (((ks * 4 + fg) ^ (pp & 7)) << 4));
        const int co = i * 16 + fr;
        bfr[i] = *reinterpret_cast<const bf16x8*>(Bs + co * 128 + (((ks * 4 + fg) ^ (co & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // D[co][pixel]
    }
  }
  lds_barrier();                                         // patch / weight stages are dead from here on
  C64_STAMP(2);

  // ---- epilogue (C tile [256 px][64 co] through LDS; pixel m = 32 * tile_row + tile_col)
  unsigned short* Cs = reinterpret_cast<unsigned short*>(smem);
  bool valid[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    valid[i] = (y0 + 2 * wave + (i >> 1) < a.H) && (x0 + (i & 1) * 16 + fr < a.W);
  auto pix_index = [&](int m) -> long long {             // flat output pixel of tile pixel m, or -1
    const int yy = y0 + (m >> 5), xx = x0 + (m & 31);
    return (yy < a.H && xx < a.W) ? (((long long)n * a.H + yy) * a.W + xx) : -1;
  };
  const int crow0 = (wave * 64 + fr) * CPAD + fg * 4;
  if (a.addend) {
    for (int idx = tid; idx < 256 * 8; idx += 256) {
      const int row = idx >> 3, ch = idx & 7;
      const long long pix = pix_index(row);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (pix >= 0) v = *reinterpret_cast<const u32x4*>(a.addend + pix * 64 + ch * 8);
      *reinterpret_cast<u32x4*>(Cs + row * CPAD + ch * 8) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(Cs + crow0 + i * 16 * CPAD + j * 16);
        acc[i][j][0] += __uint_as_float(v[0] << 16);
        acc[i][j][1] += __uint_as_float(v[0] & 0xFFFF0000u);
        acc[i][j][2] += __uint_as_float(v[1] << 16);
        acc[i][j][3] += __uint_as_float(v[1] & 0xFFFF0000u);
      }
    lds_barrier();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x2 v = {0u, 0u};                                // pixels outside the image stay exact zeros (statistics!)
      if (valid[i]) {
        v[0] = (unsigned)f32_to_bf16_bits(acc[i][j][0]) | ((unsigned)f32_to_bf16_bits(acc[i][j][1]) << 16);
        v[1] = (unsigned)f32_to_bf16_bits(acc[i][j][2]) | ((unsigned)f32_to_bf16_bits(acc[i][j][3]) << 16);
      }
      *reinterpret_cast<u32x2*>(Cs + crow0 + i * 16 * CPAD + j * 16) = v;
    }
  lds_barrier();
  if (a.stat_sum) {                                       // before the output stores: no barrier behind stores in flight
    const int col = tid & 63, part = tid >> 6;            // 4 threads per channel, 64 rows each
    float s = 0.f, q = 0.f;
#pragma unroll 16
    for (int r = part * 64; r < (part + 1) * 64; ++r) {
      const float v = bfb(Cs[r * CPAD + col]);
      s += v; q += v * v;
    }
    float* red = reinterpret_cast<float*>(smem + 256 * CPAD * 2);
    red[tid] = s; red[256 + tid] = q;
    lds_barrier();
    if (tid < 64) {
      double ds = 0.0, dq = 0.0;
#pragma unroll
      for (int p = 0; p < 4; ++p) { ds += (double)red[p * 64 + tid]; dq += (double)red[256 + p * 64 + tid]; }
      const size_t slot = (size_t)(blockIdx.x % a.stat_slots) * 64 + tid;
      atomicAdd(a.stat_sum + slot, ds);
      atomicAdd(a.stat_sumsq + slot, dq);
    }
  }
  for (int idx = tid; idx < 256 * 8; idx += 256) {
    const int row = idx >> 3, ch = idx & 7;
    const long long pix = pix_index(row);
    if (pix >= 0)
      __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(Cs + row * CPAD + ch * 8),
                                  reinterpret_cast<u32x4*>(a.out + pix * 64 + ch * 8));
  }
  C64_STAMP(3);
}


// ------------------------------------------------------------------------------------------------------------
// Persistent form: one 512-thread block per CU walks a contiguous range of tiles.
//   * the WEIGHTS LIVE IN REGISTERS: wave (pg, half) owns pixel rows 2pg, 2pg+1 of the tile and output channels
//     32*half .. +32, i.e. 9 taps x 2 k-steps x 2 MFMA tiles = 36 B-fragments = 144 VGPRs, loaded once per block;
//   * LDS holds only input patches, three of them: the DMA of tile k+2 is issued when tile k starts, so two tiles
//     of compute cover the HBM latency; ONE barrier per tile, no barrier inside it;
//   * the epilogue is register-only: the MFMA rows are permuted so that a lane ends up with 8 consecutive output
//     channels of one pixel (16 bytes): addend loads, rounding and stores go straight from / to global memory, and
//     BatchNorm statistics are accumulated in registers over ALL tiles of the block (one set of atomics per block);
//   * the two waves of a SIMD (same pixels, other channel half) are run half a phase apart: the second one defers
//     its epilogue behind the next barrier, so its VALU work runs under the first one's MFMAs and vice versa;
//   * almost no per-tile VALU work outside the epilogue: the patch image has a pitch of 40 pixels, so that a DMA
//     group (8 pixels) is (row, 8-column block) with WAVE-UNIFORM coordinates -- scalar base address + a constant
//     per-lane offset -- and only groups cut by the image border take a per-lane path.
// Loads and stores are issued in a fixed number per tile and wave, which is what the counted s_waitcnt relies on:
// pixels outside the image store into a sink, padding DMAs land in a scratch KB.
constexpr int PP = 40;                            // patch pitch in pixels (5 DMA groups per patch row)
constexpr int P_GROUPS = PH * PP / 8;             // 50 DMA groups of 8 pixels
constexpr int P_BUF = P_GROUPS * 1024;            // 51,200 B per patch
constexpr int P_NBUF = 3;
constexpr int P_SCRATCH = P_NBUF * P_BUF;         // landing zone of the padding DMAs
constexpr int P_STATS = P_SCRATCH + 1024;         // 4 pixel-group waves x 2 x 64 floats: per-wave BatchNorm partial sums
constexpr int P_LDS = P_STATS + 2048;             // 156,672 B: one block per CU
constexpr int P_DMA = 7;                          // DMA instructions per wave and tile (8 waves x 7 >= 50)

struct C64PArgs {
  const unsigned short* in;
  const unsigned short* w;
  unsigned short* out;
  const unsigned short* addend;
  const unsigned char* addend_mask;     // [pixels][8]: ReLU mask of the addend (added where its bit is set) or NULL
  double* stat_sum;
  double* stat_sumsq;
  int stat_slots;
  int N, H, W, tiles_y, tiles_x, total_tiles, tiles_per_block;
};

__device__ __attribute__((aligned(256))) unsigned char g_c64_sink[64 * 16];    // stores of out-of-image pixels
__device__ __attribute__((aligned(256))) unsigned char g_c64_zeros[2048];      // source of out-of-image DMA groups
__device__ unsigned char g_c64_ones[64] = {255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
                                           255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
                                           255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
                                           255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255};   // "no mask"

#ifdef C64_STAMPS   // per tile: wait start, barrier passed, MFMAs issued, epilogue done (wave 0 of the first 256 blocks, 32 tiles)
__device__ unsigned long long g_c64p_stamps[256 * 32 * 4];
#define C64P_STAMP(kk, k) do { if (threadIdx.x == 0 && blockIdx.x < 256 && (kk) < 32) g_c64p_stamps[(blockIdx.x * 32 + (kk)) * 4 + (k)] = clock64(); } while (0)
#else
#define C64P_STAMP(kk, k) do { } while (0)
#endif

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {     // one v_cvt_pk_bf16_f32 (round to nearest even)
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
// sum over the 16 lanes of a DPP row (the lanes of one fg group), result in every lane: row_ror 8, 4, 2, 1
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}
// LDS-DMA / store with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// (the s_nop covers the hazard "VMEM store of more than 64 bits followed by a write of its data VGPRs", which the
//  compiler cannot see through the asm statement)
// NT = false (shipped since late round 4): ordinary stores.  The two waves of a SIMD write the two 64-byte halves of a pixel
// row half a tile apart; non-temporal, each half goes to memory by itself (64-byte segments stream at 3.2 TB/s against 5.4
// for whole lines, profiles/r03_probe_rw.txt), cached, the L2 joins them: dgrad + addend 1.356 -> 1.295 ms, dgrad 0.951 ->
// 0.939 ms, forward with statistics unchanged (tools/halo_ab.py --layers l1 --exps 0,1, 4096 images).
template <bool NT = false>
__device__ __forceinline__ void store16_s(void* sbase, unsigned voff, u32x4 v) {
  if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
template <bool NT = false>
__device__ __forceinline__ void store16_v(void* ptr, u32x4 v) {
  if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}

// ABL (test entry only; results are garbage): bit 0 = no MFMAs, bit 1 = no fragment reads, bit 2 = no LDS-DMA, bit 3 = no stores
template <bool STATS, bool ADDEND, int ABL = 0>
__global__ __launch_bounds__(512) void conv3x3_c64p_kernel(C64PArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) float lds_float;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int pg = wave & 3, half = wave >> 2;             // waves w and w+4 share a SIMD: same pixels, other channel half
  const int t_begin = blockIdx.x * a.tiles_per_block;
  const int ntl = min(a.total_tiles - t_begin, a.tiles_per_block);       // >= 1 by construction of the grid
  const int tiles_img = a.tiles_y * a.tiles_x;
  lds_float* stats_lds = (lds_float*)(smem + P_STATS);

  // wave-uniform tile coordinates, advanced incrementally (no divisions in the tile loop)
  struct Tile { int n, y0, x0, pix; };                                   // pix: pixel index of the tile origin
  auto advance = [&](Tile& tl) {
    tl.x0 += TW;
    if (tl.x0 >= a.W) {
      tl.x0 = 0; tl.y0 += TH;
      if (tl.y0 >= a.H) { tl.y0 = 0; tl.n += 1; }
    }
    tl.pix = (tl.n * a.H + tl.y0) * a.W + tl.x0;
  };
  Tile cur;
  {
    const int n = t_begin / tiles_img, rem = t_begin - n * tiles_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    cur.n = n; cur.y0 = ty * TH; cur.x0 = tx * TW;
    cur.pix = (n * a.H + cur.y0) * a.W + cur.x0;
  }

  // ---- patch DMA.  Group g = wave + 8j covers patch row g/5, patch columns 8(g%5) .. +8; lane -> column offset
  //      lane/8, 16-byte slot lane%8 holding input-channel chunk (lane%8) ^ (column & 7) = (lane%8) ^ (lane/8).
  //      The group coordinates are wave-uniform and fixed: kept in scalar registers for the whole kernel.
  const int r8 = lane >> 3;
  const unsigned lane_src = (unsigned)(r8 * 128 + (((lane & 7) ^ r8) << 4));   // byte offset from the group's first pixel
  int dpy[P_DMA], dc0[P_DMA];
  unsigned doff[P_DMA], ddst[P_DMA];
#pragma unroll
  for (int j = 0; j < P_DMA; ++j) {
    const int g = wave + 8 * j;
    const int py = g / 5, gx = g - 5 * py;
    dpy[j] = g < P_GROUPS ? py : -100000;                                // padding DMA: never inside the image
    dc0[j] = 8 * gx;
    doff[j] = (unsigned)(py * a.W + 8 * gx) * 128u;                      // bytes from the patch origin
    ddst[j] = g < P_GROUPS ? (unsigned)(g * 1024) : 0xFFFFFFFFu;
  }
  // one DMA group; branch-free (per-lane address select), so that the compiler can interleave it with the MFMAs of
  // the k-step it is placed in.  live = false turns it into a padding DMA (zeros -> scratch): the NUMBER of DMAs per
  // tile never changes
  auto dma_one = [&](int j, const Tile& tl, int buf, bool live) {
    const unsigned char* origin = reinterpret_cast<const unsigned char*>(a.in) + ((long long)tl.pix - a.W - 1) * 128;
    const int yy = tl.y0 - 1 + dpy[j];                                   // wave-uniform
    const int xx = tl.x0 - 1 + dc0[j] + r8;                              // this lane's pixel column
    const bool real = live && ddst[j] != 0xFFFFFFFFu;
    const bool ok = real && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
    const unsigned char* src = (ok ? origin + doff[j] : g_c64_zeros) + lane_src;
    if (!(ABL & 4)) glds16(src, real ? lds0 + (unsigned)buf * P_BUF + ddst[j] : lds0 + P_SCRATCH);
  };
  auto issue_patch = [&](const Tile& tl, int buf, bool live) {
#pragma unroll
    for (int j = 0; j < P_DMA; ++j) dma_one(j, tl, buf, live);
  };
  Tile ahead = cur;                                                      // tile kk + 2 of the loop below
  issue_patch(ahead, 0, true);
  advance(ahead);
  issue_patch(ahead, 1, ntl > 1);
  advance(ahead);

  // ---- weights -> registers.  MFMA tile j, row rho (= fr for the A-operand fragment, fg*4+r in the result) is
  //      output channel 32*half + 8*(rho>>2) + 4*j + (rho&3): a lane's results of tiles j = 0,1 are 8 consecutive channels
  bf16x8 wreg[9][2][2];
  {
    const unsigned short* wl = a.w + (size_t)(half * 32 + (fr >> 2) * 8 + (fr & 3)) * 576 + fg * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wreg[t][ks][j] = *reinterpret_cast<const bf16x8*>(wl + (size_t)j * 4 * 576 + t * 64 + ks * 32);
  }

  // byte offset inside a patch buffer of this lane's fragment for tap column kw, k-step ks, M-tile 0, tap row 0:
  // patch pixel (row 2pg, column fr + kw); the swizzle key is the patch column & 7
  // (the second k-step is the same address with chunk bit 2 flipped: ^ 64)
  unsigned abase[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) abase[kw] = (unsigned)((2 * pg * PP + fr + kw) * 128 + ((fg ^ ((fr + kw) & 7)) << 4));
  const int co0 = half * 32 + fg * 8;                                    // first of this lane's 8 output channels
  const unsigned lane_out = (unsigned)(fr * 64 + co0) * 2;               // byte offset inside an output row segment

  f32x4 acc[4][2];
  float s8[8], q8[8];                                   // per-lane BatchNorm partial sums over ALL tiles of the block
#pragma unroll
  for (int c = 0; c < 8; ++c) { s8[c] = 0.f; q8[c] = 0.f; }

  // this lane's pixel of M-tile i: image row y0 + 2pg + (i>>1) (wave-uniform), column x0 + 16(i&1) + fr.
  // segment(i) = 0: the 16 columns are all inside the image, 1: none is (or the row is not), 2: cut by the border
  int seg_off[4];                                                        // pixel offset of M-tile i from the tile origin
#pragma unroll
  for (int i = 0; i < 4; ++i) seg_off[i] = (2 * pg + (i >> 1)) * a.W + (i & 1) * 16;
  auto segment = [&](const Tile& tl, int i, size_t& off) -> int {
    const int yy = tl.y0 + 2 * pg + (i >> 1), c0 = tl.x0 + (i & 1) * 16;
    off = (size_t)(tl.pix + seg_off[i]) * 64;
    if (yy >= a.H || c0 >= a.W) return 1;
    return c0 + 15 < a.W ? 0 : 2;
  };

  // ---- register-only epilogue of tile t.  The addend (16 bytes per M-tile) is loaded HERE, where the operand fragment
  // registers are dead, with ordinary loads: the compiler waits for them with its own vmcnt, which can only be
  // stricter than needed (it also drains the patch DMA issued before them, one tile of MFMAs earlier); the latency
  // is covered by the other wave of the SIMD, which is half a tile away.
  auto epilogue = [&](const Tile& tl) {
    u32x4 ad[4];
    unsigned amb[4];                           // the addend's ReLU-mask byte of the lane's 8 channels (0xFF: unmasked addend)
    if (ADDEND) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {            // unconditional: exactly 4 + 4 loads per tile (see CNT_TOP)
        size_t off;
        const int seg = segment(tl, i, off);
        const bool valid = seg == 0 || (seg == 2 && tl.x0 + (i & 1) * 16 + fr < a.W);
        const unsigned char* ap = valid ? reinterpret_cast<const unsigned char*>(a.addend + off) + lane_out
                                        : g_c64_zeros + lane * 16;
        ad[i] = *reinterpret_cast<const u32x4*>(ap);
        const unsigned char* mp = (valid && a.addend_mask) ? a.addend_mask + ((off * 2 + lane_out) >> 4) : g_c64_ones + lane;
        amb[i] = *mp;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      size_t off;
      const int seg = segment(tl, i, off);
      const bool valid = seg == 0 || (seg == 2 && tl.x0 + (i & 1) * 16 + fr < a.W);
      u32x4 v;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 c = acc[i][j];
        if (ADDEND) {
          const unsigned lo = ad[i][2 * j], hi = ad[i][2 * j + 1], bits = amb[i] >> (4 * j);
          c[0] += (bits & 1u) ? __uint_as_float(lo << 16) : 0.f;
          c[1] += (bits & 2u) ? __uint_as_float(lo & 0xFFFF0000u) : 0.f;
          c[2] += (bits & 4u) ? __uint_as_float(hi << 16) : 0.f;
          c[3] += (bits & 8u) ? __uint_as_float(hi & 0xFFFF0000u) : 0.f;
        }
        const unsigned w0 = pack_bf16x2(c[0], c[1]), w1 = pack_bf16x2(c[2], c[3]);
        v[2 * j] = w0;
        v[2 * j + 1] = w1;
        if (STATS && valid) {                                            // statistics of the ROUNDED outputs
          const float r0 = __uint_as_float(w0 << 16), r1 = __uint_as_float(w0 & 0xFFFF0000u);
          const float r2 = __uint_as_float(w1 << 16), r3 = __uint_as_float(w1 & 0xFFFF0000u);
          s8[4 * j + 0] += r0; q8[4 * j + 0] += r0 * r0;
          s8[4 * j + 1] += r1; q8[4 * j + 1] += r1 * r1;
          s8[4 * j + 2] += r2; q8[4 * j + 2] += r2 * r2;
          s8[4 * j + 3] += r3; q8[4 * j + 3] += r3 * r3;
        }
      }
      // exactly one store per M-tile is ISSUED whatever the validity (see CNT_TOP)
      if (ABL & 8) { asm volatile("" :: "v"(v)); }
      else if (seg == 0) store16_s<(ABL & 16) != 0>(a.out + off, lane_out, v);
      else if (seg == 1) store16_s<(ABL & 16) != 0>(g_c64_sink, (unsigned)lane * 16, v);
      else store16_v<(ABL & 16) != 0>(valid ? (void*)(reinterpret_cast<unsigned char*>(a.out + off) + lane_out) : (void*)(g_c64_sink + lane * 16), v);
    }
  };

  // In-flight vector memory operations YOUNGER than the patch DMA of tile kk (issued among the MFMAs of iteration kk-2)
  // when iteration kk starts.  The waves of channel half 0 run  [DMA kk+2 + MFMAs kk, addend loads kk, stores kk]  per
  // iteration, those of half 1 DEFER the epilogue behind the next barrier --  [addend loads kk-1, stores kk-1,
  // DMA kk+2 + MFMAs kk]  -- so that on every SIMD one wave's epilogue runs under the other wave's MFMAs.
  constexpr int NEPI = ((ABL & 8) ? 0 : 4) + (ADDEND ? 8 : 0);   // memory operations of one epilogue (stores; addend + its mask byte)
  constexpr int NDMA_T = (ABL & 4) ? 0 : P_DMA;
  constexpr int CNT_TOP0 = NEPI + NDMA_T + NEPI;        // epilogue(kk-2), DMA(kk+1), epilogue(kk-1)
  constexpr int CNT_TOP1 = NEPI + NDMA_T;               // epilogue(kk-2), DMA(kk+1)
  if (STATS) lds_barrier();                             // statistics slots zeroed

  Tile prev = cur;
  for (int kk = 0; kk < ntl; ++kk) {
    C64P_STAMP(kk, 0);
    if (kk == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (half == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT_TOP0) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT_TOP1) : "memory");
    __builtin_amdgcn_s_barrier();              // patch kk landed for every wave; buffer (kk+2)%3 is no longer read
    C64P_STAMP(kk, 1);

    if (half == 1 && kk > 0) epilogue(prev);
    const bool more = kk + 2 < ntl;
    const int buf_ahead = (kk + 2) % P_NBUF;

    // 18 k-steps (tap, half of the input channels); the fragments of step s+1 are read while step s is multiplied
    const unsigned char* patch = smem + (kk % P_NBUF) * P_BUF;
    bf16x8 af[2][4];
    auto read_step = [&](int s, bf16x8 (&f)[4]) {
      const int tap = s >> 1, ks = s & 1, kh = tap / 3, kw = tap % 3;
      const unsigned char* q = patch + (abase[kw] ^ (ks << 6));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (ABL & 2) { const unsigned u = abase[kw] + i + s; f[i] = __builtin_bit_cast(bf16x8, (u32x4){u, u, u, u}); asm volatile("" : "+v"(f[i])); }
        else f[i] = *reinterpret_cast<const bf16x8*>(q + (((i >> 1) + kh) * PP + (i & 1) * 16) * 128);
      }
    };
    read_step(0, af[0]);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      if (s + 1 < 18) read_step(s + 1, af[(s + 1) & 1]);
      if ((s & 1) && (s >> 1) < P_DMA) dma_one(s >> 1, ahead, buf_ahead, more);    // patch of tile kk+2, one group per odd step
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (ABL & 1) { if (j == 0) asm volatile("" :: "v"(af[s & 1][i])); if (s == 0) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[s >> 1][s & 1][j], af[s & 1][i],
                                                                   s == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);       // keep the fragment reads of later steps from piling up in registers
    }
    C64P_STAMP(kk, 2);
    if (half == 0) epilogue(cur);
    C64P_STAMP(kk, 3);
    prev = cur;
    advance(cur);
    advance(ahead);
  }
  if (half == 1) epilogue(prev);

  if (STATS) {
    // lanes of one fg group (a DPP row of 16) hold the same channels for different pixels; 4 pixel-group waves per half
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float s = row16_sum(s8[c]), q = row16_sum(q8[c]);
      // DETERMINISTIC (round 3): every (pixel-group wave, channel) value has ONE writer and its own LDS word; the four
      // pixel-group waves are added below in a fixed order (LDS atomics added them in arrival order: run-to-run noise
      // in the last fp32 bits of the statistics, which 17 bf16 layers amplify to per cent in the gradients)
      if (fr == 0) {
        stats_lds[pg * 128 + co0 + c] = s;
        stats_lds[pg * 128 + 64 + co0 + c] = q;
      }
    }
    lds_barrier();
    if (tid < 128) {
      const size_t slot = (size_t)(blockIdx.x % a.stat_slots) * 64 + (tid & 63);
      const float v = (stats_lds[tid] + stats_lds[128 + tid]) + (stats_lds[256 + tid] + stats_lds[384 + tid]);
      // (with stat_slots >= gridDim.x the slot row is this block's own: 0 + v is exact, bit-reproducible)
      atomicAdd((tid < 64 ? a.stat_sum : a.stat_sumsq) + slot, (double)v);
    }
  }
}

template <bool STATS, bool ADDEND, int ABL = 0>
int launch_c64p(const C64PArgs& a, int grid, hipStream_t stream) {
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (isic_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64p_kernel<STATS, ADDEND, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  P_LDS);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL((conv3x3_c64p_kernel<STATS, ADDEND, ABL>), dim3(grid), dim3(512), P_LDS, stream, a);
  return ISIC_OK;
}

}  // namespace

// called by isic_conv2d_igemm_bf16 for Cin = Cout = 64, 3x3, stride 1, pad 1; variant 1 = one tile per block,
// variant 2 = persistent blocks with register-resident weights
int isic_conv3x3_c64_launch(int variant, const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int H, int W,
                            const uint16_t* addend, const uint8_t* addend_mask, double* stat_sum, double* stat_sumsq,
                            int stat_slots, int experiment, hipStream_t stream) {
  const int tiles_y = ceil_div(H, TH), tiles_x = ceil_div(W, TW);
  const int64_t blocks = (int64_t)N * tiles_y * tiles_x;
  if (blocks > 0x7FFFFFFFLL) return ISIC_ERR_UNSUPPORTED;
  if (variant == 2 && !(stat_sum && addend)) {
    const int cus = isic_cu_count();
    C64PArgs a;
    a.in = in; a.w = w; a.out = out; a.addend = addend; a.addend_mask = addend_mask;
    a.stat_sum = stat_sum; a.stat_sumsq = stat_sumsq; a.stat_slots = stat_slots > 0 ? stat_slots : 1;
    a.N = N; a.H = H; a.W = W; a.tiles_y = tiles_y; a.tiles_x = tiles_x;
    a.total_tiles = (int)blocks;
    a.tiles_per_block = ceil_div(a.total_tiles, cus);
    const int grid = ceil_div(a.total_tiles, a.tiles_per_block);       // every block owns at least one tile
    if (experiment == 1) {                                 // A/B: the non-temporal stores of rounds 2-3 instead of ordinary ones
      if (stat_sum) return launch_c64p<true, false, 16>(a, grid, stream);
      if (addend) return launch_c64p<false, true, 16>(a, grid, stream);
      return launch_c64p<false, false, 16>(a, grid, stream);
    }
    if (experiment >= 2) {                                 // timing ablations: experiment digit e -> where the time goes
      if (stat_sum || addend) return ISIC_ERR_UNSUPPORTED;
      switch (experiment) {
        case 2: return launch_c64p<false, false, 1>(a, grid, stream);     // no MFMAs
        case 3: return launch_c64p<false, false, 8>(a, grid, stream);     // no stores
        case 4: return launch_c64p<false, false, 4>(a, grid, stream);     // no DMA
        case 5: return launch_c64p<false, false, 12>(a, grid, stream);    // no DMA, no stores: MFMA + fragment reads
        case 6: return launch_c64p<false, false, 11>(a, grid, stream);    // DMA only
        case 7: return launch_c64p<false, false, 7>(a, grid, stream);     // stores only
        default: return ISIC_ERR_UNSUPPORTED;
      }
    }
    if (stat_sum) return launch_c64p<true, false>(a, grid, stream);
    if (addend) return launch_c64p<false, true>(a, grid, stream);
    return launch_c64p<false, false>(a, grid, stream);
  }
  if (addend_mask) return ISIC_ERR_UNSUPPORTED;            // the persistent kernel above is the one that takes a masked addend
  C64Args a;
  a.in = in; a.w = w; a.out = out; a.addend = addend;
  a.stat_sum = stat_sum; a.stat_sumsq = stat_sumsq; a.stat_slots = stat_slots > 0 ? stat_slots : 1;
  a.N = N; a.H = H; a.W = W;
  a.tiles_y = tiles_y; a.tiles_x = tiles_x;
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (isic_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_BYTES);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL(conv3x3_c64_kernel, dim3((unsigned)blocks), dim3(256), LDS_BYTES, stream, a);
  return ISIC_OK;
}
