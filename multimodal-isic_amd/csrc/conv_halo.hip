// 3x3 / stride 1 / pad 1 convolution for the >= 128-channel layers with the INPUT PIXELS STAGED ONCE FOR ALL NINE
// TAPS (gfx950, bf16 MFMA): ResNet-18 layer2..4 forward and, with flipped weights, data gradient.
//
// The generic implicit GEMM (conv_igemm.hip) stages a 256-pixel x 64-channel A tile once PER TAP: 9 x 32 KB of
// activations next to 9 x 16 KB of weights per 64 input channels and 128 output channels, 85 FLOP per staged byte --
// and the chip delivers only ~10 TB/s of L2 -> LDS staging (profiles/r01_*: the kernel sits at 0.30-0.37 of the MFMA
// peak while drawing that rate).  Here a tile is 256 CONSECUTIVE pixels of the flattened (n, h, w) index space and
// the block stages the pixel rows [m0 - (W+1), m0 + 256 + (W+1)) of a 64-channel chunk ONCE (40 KB at W = 28): tap
// (kh, kw) of local pixel r is patch row r + kh*W + kw, whatever image rows the tile crosses.  Taps that fall outside
// the image (pad) are not zero-filled in the patch -- the same patch row is a valid neighbour for one pixel and a
// padded one for another -- but masked at the fragment read: an invalid lane reads a 128-byte block of zeros instead.
// Staged bytes per 64 input channels: 40 KB + 9 x 16 KB = 184 KB instead of 432 KB (200 FLOP per staged byte).
//
// Structure (one persistent 1024-thread block per CU):
//   * waves 0-7 multiply: 4 (pixels) x 2 (channels) waves of 64 x 64 = 4 x 4 v_mfma_f32_16x16x32_bf16 tiles, operand
//     roles swapped as in conv_igemm.hip (a lane ends up with 4 consecutive output channels of one pixel);
//   * waves 8-15 only issue LDS-DMA (global_load_lds, 16 B per lane): the weights of K-tile (tap, chunk) + 2 into a
//     three-stage ring, and the NEXT chunk's patch -- also the next tile's first chunk -- one piece per tap into the
//     second patch buffer, so that nothing but the very first patch of a block is exposed;
//   * one s_barrier per K-tile; the staging waves count their own DMAs (exactly three per K-tile each, padded with
//     dummy pieces), s_waitcnt vmcnt(3) == "this K-tile's weights and everything older have landed";
//   * register-only epilogue: the weight rows are permuted inside a stage so that a lane ends up with EIGHT consecutive
//     output channels (16-byte stores, 64 contiguous bytes per pixel and instruction); fused residual-gradient addend
//     (all loads before the first store, one rounding) and BatchNorm sum / sum of squares of the rounded outputs,
//     summed per block in LDS and flushed once (fp64 atomics into `stat_slots` rows).
//   A block keeps one 128-channel output slice for its whole life (blocks are dealt over the Cout/128 slices).
//
// The reference has no convolution kernel of its own (its encoder is an un-vendored ConvMAE run through torch,
// save_latent.py:42-60); BASELINE.json configs[1] names ResNet-18 (SURVEY.md 8d layer table).
#include <type_traits>

#include "common.h"

namespace {

constexpr int HM = 256;                 // pixels per tile
constexpr int HN = 128;                 // output channels per block
constexpr int WSTAGE = HN * 128;        // one (tap, 64-channel chunk) of weights: 128 rows x 128 B
constexpr int NWST = 3;
constexpr int MAX_PROWS = 384;          // W <= 63
constexpr int MAX_PPW = MAX_PROWS / 64; // patch pieces (1 KB DMA groups) per staging wave and chunk

struct HaloArgs {
  const unsigned short* in;
  const unsigned short* w;       // [Cout][3][3][Cin]
  unsigned short* out;
  const unsigned short* addend;
  double* stat_sum;              // STATS 1: sum of the outputs;  STATS 2: sum of dz
  double* stat_sumsq;            // STATS 1: sum of squares;      STATS 2: sum of dz * y
  const unsigned char* addend_mask; // STATS 0 + addend: [M][Cout/8] ReLU mask of the ADDEND (it is added where its bit is set) or NULL
  const unsigned char* relu_mask;   // STATS 2: [M][Cout/8], bit c of a byte = ReLU mask of channel 8g + c (isic_bn_apply_mask_bf16)
  const unsigned short* yraw;       // STATS 2: [M][Cout] the pre-BatchNorm activation of the layer whose gradient this is
  int stat_slots;
  int H, W, Cin, Cout, M;        // M = N*H*W
  int mtiles, nslices, groups;   // ceil(M/256), Cout/128, blocks per slice
  int tiles_per_block, cchunks, prows;
  unsigned long long magic_hw, magic_w;   // floor(2^40/d)+1 for d = H*W and d = W (M < 2^24)
};

__device__ __attribute__((aligned(256))) unsigned char g_halo_zero_page[256];

__device__ __forceinline__ unsigned fdiv40(unsigned n, unsigned long long magic) {
  return (unsigned)(((unsigned long long)n * magic) >> 40);
}
// 16-byte-per-lane LDS-DMA in inline asm: outside hipcc's vmcnt bookkeeping (the staging waves count by hand).
__device__ __forceinline__ void halo_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
typedef __attribute__((ext_vector_type(2))) float hf32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 hbf16x2;
__device__ __forceinline__ unsigned halo_pack2(float lo, float hi) {     // one v_cvt_pk_bf16_f32 (round to nearest even)
  const hf32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, hbf16x2));
}
__device__ __forceinline__ float halo_row16_sum(float v) {               // sum over the 16 lanes of a DPP row
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

#ifdef HALO_STAMPS   // tests/probes/probe_halo_stamps.hip: per-tile phase timestamps of MFMA wave 0 of the first 256 blocks
__device__ unsigned long long g_halo_stamps[256 * 64 * 4];
#define HALO_STAMP(tl, k) do { if (threadIdx.x == 0 && blockIdx.x < 256 && (tl) < 64) g_halo_stamps[(blockIdx.x * 64 + (tl)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HALO_STAMP(tl, k) do { } while (0)
#endif

// LDS map (bytes): [0, 2*PB) two patch buffers (PB = prows*128) | 3 weight stages | 128 B zeros | 1 KB DMA scratch |
// statistics: 8 MFMA waves x 128 floats (each wave's own slots)
//
// STATS: 0 none; 1 forward -- BatchNorm sum / sum of squares of the rounded outputs; 2 DATA GRADIENT FEEDING A
// BatchNorm(+ReLU) BACKWARD -- the output g (+ addend) is the gradient of a ReLU(BatchNorm(y)) activation: the epilogue
// applies the ReLU mask (1 bit per value, written by the forward), stores dz = mask ? g : 0 and accumulates the two
// sums BatchNorm's backward needs, sum dz and sum dz * y, per channel -- the separate reduction pass over (g, y) and,
// for a block's last BatchNorm, the materialised residual gradient (= dz) disappear.
//
// KPB = K-tiles per barrier.  1: the three-stage weight ring above.  2: FOUR weight stages and one barrier per TWO K-tiles
// (64 MFMAs per wave between barriers): after barrier S the staging waves fetch the two K-tiles of step S + 1 into the
// two stages step S - 1 used and wait for ALL their own DMAs (vmcnt(0): no dummy pieces) in front of barrier S + 1.
// The next chunk's patch pieces go out at taps 1..6 (not 0..5): a step may hold (chunk c - 1, tap 8) and (chunk c, tap 0),
// and the buffer of chunk c + 1 is the one chunk c - 1 reads.  Needs Cin % 128 == 0 (an even number of K-tiles per tile).
// ABL (test entry only): bit 0 = no MFMAs, bit 1 = no fragment reads, bit 2 = no LDS-DMA -- where a tile's time goes
template <int STATS, bool ADDEND, int KPB, int PRIO, int ABL = 0>
__global__ __launch_bounds__(1024) void conv_halo_kernel(HaloArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  constexpr int NST = KPB == 2 ? 4 : NWST;
  const int PB = a.prows * 128;
  const int off_w = 2 * PB, off_zero = off_w + NST * WSTAGE, off_scr = off_zero + 128, off_stat = off_scr + 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // block -> (output slice, contiguous range of pixel tiles)
  const int slice = blockIdx.x % a.nslices, grp = blockIdx.x / a.nslices;
  const int t_begin = grp * a.tiles_per_block;
  const int ntl = min(a.mtiles - t_begin, a.tiles_per_block);
  if (ntl <= 0) return;                                  // whole block: no barrier has been reached yet
  const int n0 = slice * HN;
  const int HWs = a.H * a.W;

  typedef __attribute__((address_space(3))) float lds_float;
  lds_float* stats_lds = (lds_float*)(smem + off_stat);
  if (wave >= 8) {
    // =================================================================== staging waves
    const int sw = wave - 8;
    const int r8 = lane >> 3;
    const int gch = (lane & 7) ^ r8;                     // global 16-byte chunk this lane fetches (swizzle on the source)
    const unsigned char* zp = g_halo_zero_page + (lane & 7) * 16;
    const int groups = a.prows >> 3;
    const int total_chunks = ntl * a.cchunks;
    const unsigned scr = lds0 + off_scr;

    // one 1 KB piece of the patch of (tile index tl, chunk cc): group g = sw + 8*j, patch rows 8g .. 8g+7
    auto patch_piece = [&](int j, int tl, int cc, int buf, bool live) {
      const int g = sw + 8 * j;
      const bool real = live && g < groups;
      const long long pix = (long long)(t_begin + tl) * HM - (a.W + 1) + g * 8 + r8;
      const bool ok = real && pix >= 0 && pix < a.M;
      const void* src = ok ? (const void*)(a.in + (size_t)pix * a.Cin + cc * 64 + gch * 8) : (const void*)zp;
      halo_glds16(src, real ? lds0 + (unsigned)(buf * PB + g * 1024) : scr);
    };
    // weights of K-tile (tap, cc): stage rows sr = 16*sw + 8*t + r8 (t = 0, 1: the wave's two DMA groups).  Stage row
    // wn*64 + j*16 + rho feeds row rho of MFMA tile j of channel half wn and holds output channel
    //   wn*64 + 32*(j>>1) + 8*(rho>>2) + 4*(j&1) + (rho&3)
    // so that a lane's results of tiles 2t, 2t+1 are EIGHT CONSECUTIVE channels (one 16-byte store per pixel).
    const int chan0 = (sw >> 2) * 64 + ((sw & 3) >> 1) * 32 + (r8 >> 2) * 8 + (sw & 1) * 4 + (r8 & 3);   // t = 0; t = 1: + 16
    const unsigned short* wrow = a.w + ((size_t)(n0 + chan0) * 9) * a.Cin + gch * 8;
    auto weights = [&](int tap, int cc, int stage, bool live) {
      const unsigned short* s0 = wrow + (size_t)tap * a.Cin + cc * 64;
      const unsigned dst = lds0 + off_w + stage * WSTAGE + sw * 2048;
      halo_glds16(live ? (const void*)s0 : (const void*)zp, live ? dst : scr);
      halo_glds16(live ? (const void*)(s0 + (size_t)16 * 9 * a.Cin) : (const void*)zp, live ? dst + 1024 : scr);
    };

    if constexpr (KPB == 2) {
      // ---- one barrier per two K-tiles, four weight stages, every DMA of a step waited for before the next barrier
      auto piece2 = [&](int j, int tl, int cc, int buf) {    // as patch_piece, but nothing is issued for a dead piece
        const int g = sw + 8 * j;
        if (g >= groups) return;
        const long long pix = (long long)(t_begin + tl) * HM - (a.W + 1) + g * 8 + r8;
        const bool ok = pix >= 0 && pix < a.M;
        if (!(ABL & 4))
          halo_glds16(ok ? (const void*)(a.in + (size_t)pix * a.Cin + cc * 64 + gch * 8) : (const void*)zp,
                      lds0 + (unsigned)(buf * PB + g * 1024));
      };
      auto weights2 = [&](int tap, int cc, int stage) {
        const unsigned short* s0 = wrow + (size_t)tap * a.Cin + cc * 64;
        const unsigned dst = lds0 + off_w + stage * WSTAGE + sw * 2048;
        if (!(ABL & 4)) {
          halo_glds16((const void*)s0, dst);
          halo_glds16((const void*)(s0 + (size_t)16 * 9 * a.Cin), dst + 1024);
        }
      };
#pragma unroll
      for (int j = 0; j < MAX_PPW; ++j) piece2(j, 0, 0, 0);
      weights2(0, 0, 0);
      weights2(1, 0, 1);
      const int total_kt = total_chunks * 9;                 // even (Cin % 128 == 0)
      // (tap, chunk, tile, chunk counter) of the K-tile whose patch pieces are issued: the CURRENT step's K-tiles;
      // (tap, chunk) of the K-tile whose weights are issued: two K-tiles ahead
      int ptap = 0, pcc = 0, ptl = 0, pgc = 0;
      int wtap = 2, wcc = 0;
      for (int g = 0; g < total_kt; g += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (g + 2 + h < total_kt) weights2(wtap, wcc, (g + 2 + h) & 3);
          if (++wtap == 9) { wtap = 0; if (++wcc == a.cchunks) wcc = 0; }
          if (ptap >= 1 && ptap <= MAX_PPW && pgc + 1 < total_chunks) {
            const bool last = pcc + 1 == a.cchunks;
            piece2(ptap - 1, last ? ptl + 1 : ptl, last ? 0 : pcc + 1, (pgc + 1) & 1);
          }
          if (++ptap == 9) { ptap = 0; ++pgc; if (++pcc == a.cchunks) { pcc = 0; ++ptl; } }
        }
      }
    } else {
#pragma unroll
    for (int j = 0; j < MAX_PPW; ++j) patch_piece(j, 0, 0, 0, true);
    weights(0, 0, 0, true);
    weights(1, 0, 1, true);                              // a chunk always has 9 K-tiles: K-tile 1 exists

    int gc = 0;                                          // chunk counter over (tile, chunk)
    for (int tl = 0; tl < ntl; ++tl) {
      for (int cc = 0; cc < a.cchunks; ++cc, ++gc) {
        const bool more_chunks = gc + 1 < total_chunks;
        const int ncc = cc + 1 == a.cchunks ? 0 : cc + 1;
        const int ntl_ = cc + 1 == a.cchunks ? tl + 1 : tl;
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int tap = kh * 3 + kw;                 // weight stage of K-tile (tap, chunk) = tap % 3 = kw
            // this K-tile's weights (issued two K-tiles ago) and everything older -- the patch pieces of this chunk
            // among it -- have landed when only the previous K-tile's three DMAs are still in flight
            if (gc == 0 && tap == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            __builtin_amdgcn_s_barrier();    // ... for every staging wave; the MFMA waves are done with K-tile it-1
            // weights of K-tile it+2 into the stage K-tile it-1 used
            const bool wrap = tap + 2 >= 9;
            const int t2 = wrap ? tap + 2 - 9 : tap + 2;
            weights(t2, wrap ? ncc : cc, (kw + 2) % NWST, !wrap || more_chunks);
            // one piece of the next chunk's patch into the buffer the previous chunk used
            patch_piece(tap, ntl_, ncc, (gc + 1) & 1, more_chunks && tap < MAX_PPW);
          }
        }
      }
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may outlive the block's LDS allocation
  } else {
  // ===================================================================== MFMA waves
  if (tid < 8) reinterpret_cast<u32x4*>(smem + off_zero)[tid] = (u32x4){0u, 0u, 0u, 0u};   // 128 B of zeros
  if (STATS != 0 && tid < 512) { stats_lds[tid] = 0.f; stats_lds[tid + 512] = 0.f; }   // 8 waves x 2 x 64
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // ordered before the first barrier of the K loop

  const int fr = lane & 15, fg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  // weight fragment byte offsets inside a stage: row wn*64 + j*16 + fr, chunk fg (k-step 0; k-step 1 = ^ 64)
  // (rows 16 apart share the swizzle key row & 7: tile j / i is the tile-0 address + 2048 j / i)
  const unsigned boff0 = (unsigned)((wn * 64 + fr) * 128 + ((fg ^ (fr & 7)) << 4));
  const int prow0 = wm * 64 + fr;                        // patch row of tap (0, 0) of this lane's first pixel

  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);             // the multiplying waves win vector-issue arbitration against the staging waves
  int gc = 0, gkt = 0;
  for (int tl = 0; tl < ntl; ++tl) {
    const int m0 = (t_begin + tl) * HM;
    // this lane's four pixels: tap validity (bit tap) and the patch row of tap (0, 0)
    unsigned vmask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wm * 64 + i * 16 + fr;
      const int m = m0 + r;
      unsigned vm = 0u;
      if (m < a.M) {
        const int n = (int)fdiv40((unsigned)m, a.magic_hw);
        const int rem = m - n * HWs;
        const int h = (int)fdiv40((unsigned)rem, a.magic_w), w = rem - h * a.W;
        const unsigned vh = (h >= 1 ? 1u : 0u) | 2u | (h + 1 < a.H ? 4u : 0u);
        const unsigned vw = (w >= 1 ? 1u : 0u) | 2u | (w + 1 < a.W ? 4u : 0u);
#pragma unroll
        for (int t = 0; t < 9; ++t) vm |= (((vh >> (t / 3)) & (vw >> (t % 3))) & 1u) << t;
      }
      vmask[i] = vm;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    HALO_STAMP(tl, 0);                                     // tile set-up done, K loop starts

    if constexpr (KPB == 2) {
      int tap = 0, kh = 0, kw = 0;
      const int nkt = a.cchunks * 9;
#pragma unroll 1
      for (int kt = 0; kt < nkt; kt += 2) {
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {                      // (not unrolled: hipcc otherwise hoists the second K-tile's fragment
          const int pbase = (gc & 1) * PB;                 //  reads over the first one's multiplies and spills)
          const unsigned char* wst = smem + off_w + ((gkt + h) & 3) * WSTAGE;
          const int toff = kh * a.W + kw;
          unsigned aoff[4];
          const int p = prow0 + toff;
          const int real0 = pbase + p * 128 + ((fg ^ (p & 7)) << 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)(((vmask[i] >> tap) & 1u) ? real0 + i * 2048 : off_zero);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (ABL & 2) {
                const unsigned u = aoff[i] + ks, v = boff0 + i;
                af[i] = __builtin_bit_cast(bf16x8, (u32x4){u, u, u, u});
                bfr[i] = __builtin_bit_cast(bf16x8, (u32x4){v, v, v, v});
                asm volatile("" : "+v"(af[i]), "+v"(bfr[i]));
              } else {
                af[i] = *reinterpret_cast<const bf16x8*>(smem + (aoff[i] ^ (unsigned)(ks << 6)));
                bfr[i] = *reinterpret_cast<const bf16x8*>(wst + ((boff0 ^ (unsigned)(ks << 6)) + i * 2048));
              }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (ABL & 1) { if (j == 0) asm volatile("" :: "v"(af[i]), "v"(bfr[i])); }
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
              }
          }
          ++tap;
          if (++kw == 3) { kw = 0; ++kh; }
          if (tap == 9) { tap = 0; kh = 0; ++gc; }
        }
        gkt += 2;
      }
    } else {
    for (int cc = 0; cc < a.cchunks; ++cc, ++gc) {
      const int pbase = (gc & 1) * PB;
#pragma unroll 1
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll 1
        for (int kw = 0; kw < 3; ++kw) {
          __builtin_amdgcn_s_barrier();
          const unsigned char* wst = smem + off_w + kw * WSTAGE;     // stage of K-tile (tap, chunk) = tap % 3 = kw
          const int tap = kh * 3 + kw;
          const int toff = kh * a.W + kw;
          unsigned aoff[4];                                // byte offset of the k-step-0 fragment (k-step 1: ^ 64)
          const int p = prow0 + toff;
          const int real0 = pbase + p * 128 + ((fg ^ (p & 7)) << 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)(((vmask[i] >> tap) & 1u) ? real0 + i * 2048 : off_zero);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              af[i] = *reinterpret_cast<const bf16x8*>(smem + (aoff[i] ^ (unsigned)(ks << 6)));
              bfr[i] = *reinterpret_cast<const bf16x8*>(wst + ((boff0 ^ (unsigned)(ks << 6)) + i * 2048));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
          }
        }
      }
    }

    }
    HALO_STAMP(tl, 1);                                     // K loop issued
    // ---- register-only epilogue: lane (fg, fr) holds, for MFMA tiles (i, 2t) and (i, 2t+1), the eight consecutive
    //      output channels n0 + wn*64 + 32t + 8fg + {0..7} of pixel m0 + wm*64 + i*16 + fr: one 16-byte access each.
    //      Every addend load is issued before the first store (vmcnt counts loads and stores in order: a load behind
    //      a store would wait for the store's round trip), and nothing behind the stores waits on vmcnt.
    const unsigned chan = (unsigned)(n0 + wn * 64 + fg * 8);
    if constexpr (STATS == 2) {
      // One pixel tile i at a time with TWO tiles' operands (addend, y, mask byte) in flight: the loads of tile i + 2
      // are issued before the stores of tile i (a load behind a store would wait for the store's round trip).  No sums
      // are carried across tiles in registers -- that is what makes room for the second set of operands: each tile's
      // 16 values per lane are reduced over the DPP row at once and added to the wave's own LDS slots.
      // DETERMINISTIC sums (they feed back into the gradient chain, which amplifies any run-to-run difference): no
      // atomics, fixed order; the block's flush adds the four waves of a channel half in a fixed order.
      u32x4 ad[2], yv[2][2];                                 // addend: one tile ahead; y and mask: two tiles ahead
      unsigned mb[2][2];
      // (the tile index is a compile-time constant INSIDE the lambda: with a run-time parameter the operand arrays are
      //  indexed dynamically before inlining and end up in scratch memory)
      auto load_tile = [&](auto IC) {
        constexpr int i = decltype(IC)::value;
        const int m = m0 + wm * 64 + i * 16 + fr;
        const unsigned off = (unsigned)(m < a.M ? m : 0) * (unsigned)a.Cout + chan;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          yv[i & 1][t] = *reinterpret_cast<const u32x4*>(a.yraw + off + t * 32);
          mb[i & 1][t] = a.relu_mask[(off + t * 32) >> 3];
        }
      };
      auto load_addend = [&](int i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        const unsigned off = (unsigned)(m < a.M ? m : 0) * (unsigned)a.Cout + chan;
#pragma unroll
        for (int t = 0; t < 2; ++t) ad[t] = *reinterpret_cast<const u32x4*>(a.addend + off + t * 32);
      };
      if (ADDEND) load_addend(0);
      load_tile(std::integral_constant<int, 0>{});
      load_tile(std::integral_constant<int, 1>{});
      lds_float* sp = stats_lds + wave * 128 + lane;
      float tot[2] = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        const bool valid = m < a.M;
        const unsigned off = (unsigned)(valid ? m : 0) * (unsigned)a.Cout + chan;
        u32x4 res[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const unsigned bits = valid ? mb[i & 1][t] : 0u;
          float s8[8], q8[8];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x4 c = acc[i][2 * t + h];
            if (ADDEND) {
              const unsigned lo = ad[t][2 * h], hi = ad[t][2 * h + 1];
              c[0] += __uint_as_float(lo << 16);
              c[1] += __uint_as_float(lo & 0xFFFF0000u);
              c[2] += __uint_as_float(hi << 16);
              c[3] += __uint_as_float(hi & 0xFFFF0000u);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = ((bits >> (4 * h + e)) & 1u) ? c[e] : 0.f;
            const unsigned w0 = halo_pack2(c[0], c[1]), w1 = halo_pack2(c[2], c[3]);
            res[t][2 * h] = w0;
            res[t][2 * h + 1] = w1;
            const unsigned ylo = yv[i & 1][t][2 * h], yhi = yv[i & 1][t][2 * h + 1];
            const float r0 = __uint_as_float(w0 << 16), r1 = __uint_as_float(w0 & 0xFFFF0000u);       // the ROUNDED dz
            const float r2 = __uint_as_float(w1 << 16), r3 = __uint_as_float(w1 & 0xFFFF0000u);
            s8[4 * h + 0] = r0; q8[4 * h + 0] = r0 * __uint_as_float(ylo << 16);
            s8[4 * h + 1] = r1; q8[4 * h + 1] = r1 * __uint_as_float(ylo & 0xFFFF0000u);
            s8[4 * h + 2] = r2; q8[4 * h + 2] = r2 * __uint_as_float(yhi << 16);
            s8[4 * h + 3] = r3; q8[4 * h + 3] = r3 * __uint_as_float(yhi & 0xFFFF0000u);
          }
          float mine = 0.f;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float sv = halo_row16_sum(s8[c]), qv = halo_row16_sum(q8[c]);
            mine = fr == c ? sv : mine;
            mine = fr == 8 + c ? qv : mine;
          }
          tot[t] += mine;                                    // fixed order i = 0..3
          asm volatile("" : "+v"(tot[t]));                   // computed HERE: hipcc otherwise sinks all 128 partial sums
                                                             // of a tile to its end -- and spills them
        }
        if (ADDEND && i + 1 < 4) load_addend(i + 1);         // (every load of the epilogue is issued before a store it follows)
        if (i == 0) load_tile(std::integral_constant<int, 2>{});      // before this tile's stores
        if (i == 1) load_tile(std::integral_constant<int, 3>{});
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (valid) __builtin_nontemporal_store(res[t], reinterpret_cast<u32x4*>(a.out + off + t * 32));
        __builtin_amdgcn_sched_barrier(0);                   // one tile at a time: keeps the register footprint bounded
      }
      sp[0] += tot[0];
      sp[64] += tot[1];
      continue;
    }
    u32x4 ad[4][2];
    unsigned amb[4][2];              // the addend's ReLU-mask byte (8 channels); 0xFF when the addend comes unmasked
    if (ADDEND) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        const unsigned off = (unsigned)(m < a.M ? m : 0) * (unsigned)a.Cout + chan;     // M * Cout < 2^31 (host check)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          ad[i][t] = *reinterpret_cast<const u32x4*>(a.addend + off + t * 32);
          amb[i][t] = a.addend_mask ? (unsigned)a.addend_mask[(off + t * 32) >> 3] : 0xFFu;
        }
      }
    }
    float s8[2][8], q8[2][8];
    if (STATS == 1) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 8; ++c) { s8[t][c] = 0.f; q8[t][c] = 0.f; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      const bool valid = m < a.M;
      const unsigned off = (unsigned)(valid ? m : 0) * (unsigned)a.Cout + chan;
      u32x4 vv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        u32x4 v;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 c = acc[i][2 * t + h];
          if (ADDEND) {
            const unsigned lo = ad[i][t][2 * h], hi = ad[i][t][2 * h + 1], bits = amb[i][t] >> (4 * h);
            c[0] += (bits & 1u) ? __uint_as_float(lo << 16) : 0.f;
            c[1] += (bits & 2u) ? __uint_as_float(lo & 0xFFFF0000u) : 0.f;
            c[2] += (bits & 4u) ? __uint_as_float(hi << 16) : 0.f;
            c[3] += (bits & 8u) ? __uint_as_float(hi & 0xFFFF0000u) : 0.f;
          }
          const unsigned w0 = halo_pack2(c[0], c[1]), w1 = halo_pack2(c[2], c[3]);
          v[2 * h] = w0;
          v[2 * h + 1] = w1;
          if (STATS == 1 && valid) {                     // statistics of the ROUNDED outputs
            const float r0 = __uint_as_float(w0 << 16), r1 = __uint_as_float(w0 & 0xFFFF0000u);
            const float r2 = __uint_as_float(w1 << 16), r3 = __uint_as_float(w1 & 0xFFFF0000u);
            s8[t][4 * h + 0] += r0; q8[t][4 * h + 0] += r0 * r0;
            s8[t][4 * h + 1] += r1; q8[t][4 * h + 1] += r1 * r1;
            s8[t][4 * h + 2] += r2; q8[t][4 * h + 2] += r2 * r2;
            s8[t][4 * h + 3] += r3; q8[t][4 * h + 3] += r3 * r3;
          }
        }
        vv[t] = v;
      }
      // whole 128-byte lines per store instruction (common.h: isic_pair_rows): 8 rows x 128 B instead of 16 rows x 64 B
      {
        const bool odd = fr & 1;
        u32x4 da, db;
        isic_pair_rows(vv[0], vv[1], odd, da, db);
        const int mA = m0 + wm * 64 + i * 16 + (fr & ~1), mB = mA + 1;
        const unsigned col = chan + (odd ? 32u : 0u);
        if (mA < a.M) __builtin_nontemporal_store(da, reinterpret_cast<u32x4*>(a.out + (unsigned)mA * (unsigned)a.Cout + col));
        if (mB < a.M) __builtin_nontemporal_store(db, reinterpret_cast<u32x4*>(a.out + (unsigned)mB * (unsigned)a.Cout + col));
      }
    }
    if (STATS == 1) {
      // lanes of one fg group (a DPP row of 16) hold the same 8 channels for 16 different pixels: every lane ends with
      // the row totals of the 16 values (8 sums, 8 sums of squares) of a channel group, and lane fr keeps value #fr.
      // DETERMINISTIC (round 3): a wave adds its tiles' values to its OWN LDS slots in tile order -- no LDS atomics, whose
      // arrival order between the four wm waves changed the fp32 sums from run to run; the block's flush adds the four
      // waves of a channel half in a fixed order (the STATS 2 scheme), and every block owns its own fp64 slot row.
      lds_float* sp = stats_lds + wave * 128 + lane;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float mine = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float sv = halo_row16_sum(s8[t][c]), qv = halo_row16_sum(q8[t][c]);
          mine = fr == c ? sv : mine;
          mine = fr == 8 + c ? qv : mine;
        }
        sp[t * 64] += mine;
      }
    }
    HALO_STAMP(tl, 2);                                     // epilogue issued
  }
  }   // MFMA waves

  if (STATS != 0) {
    lds_barrier();                                       // all sixteen waves: every tile's partial sums are in LDS
    if (tid < 256) {
      const size_t slot = (size_t)(blockIdx.x % a.stat_slots) * a.Cout + n0 + (tid & 127);
      // channel c of the slice = wn*64 + t*32 + fg*8 + e lives in lane fg*16 + e (+ 8 for the second sum) of the
      // slots of waves wm*2 + wn, wm = 0..3
      const int c = tid & 127, l = ((c >> 3) & 3) * 16 + (c & 7) + (tid < 128 ? 0 : 8), t = (c >> 5) & 1, wn_ = c >> 6;
      float v = 0.f;
#pragma unroll
      for (int wm_ = 0; wm_ < 4; ++wm_) v += stats_lds[(wm_ * 2 + wn_) * 128 + t * 64 + l];
      // (an atomic only because a caller MAY pass fewer slot rows than blocks; with stat_slots >= gridDim.x the row is
      //  this block's own and 0 + v is exact: bit-reproducible)
      atomicAdd((tid < 128 ? a.stat_sum : a.stat_sumsq) + slot, (double)v);
    }
  }
}

template <int STATS, bool ADDEND, int KPB, int PRIO, int ABL = 0>
int launch_halo(const HaloArgs& a, int grid, int lds, hipStream_t stream) {
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (isic_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<STATS, ADDEND, KPB, PRIO, ABL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL((conv_halo_kernel<STATS, ADDEND, KPB, PRIO, ABL>), dim3(grid), dim3(1024), lds, stream, a);
  return ISIC_OK;
}

}  // namespace

// 3x3 / stride 1 / pad 1, Cin % 64 == 0, Cout % 128 == 0, W <= 63: called by isic_conv2d_igemm_bf16 (conv_igemm.hip).
bool isic_conv_halo_supported(int N, int H, int W, int Cin, int Cout) {
  const long long M = (long long)N * H * W;
  return Cin % 64 == 0 && Cout % HN == 0 && W >= 1 && 256 + 2 * W + 2 <= MAX_PROWS && M < (1LL << 24) && H * W < (1 << 16) &&
         M * Cin <= 0x7FFFFFFFLL && M * Cout <= 0x7FFFFFFFLL;
}

int isic_conv_halo_launch(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int H, int W, int Cin, int Cout,
                          const uint16_t* addend, const uint8_t* addend_mask, double* stat_sum, double* stat_sumsq,
                          int stat_slots, const uint8_t* relu_mask, const uint16_t* yraw, int experiment, hipStream_t stream) {
  if (!isic_conv_halo_supported(N, H, W, Cin, Cout)) return ISIC_ERR_UNSUPPORTED;
  const int cus = isic_cu_count();
  HaloArgs a;
  a.in = in; a.w = w; a.out = out; a.addend = addend; a.addend_mask = addend_mask;
  a.stat_sum = stat_sum; a.stat_sumsq = stat_sumsq; a.stat_slots = stat_slots > 0 ? stat_slots : 1;
  a.relu_mask = relu_mask; a.yraw = yraw;
  if (addend_mask && (!addend || relu_mask || yraw || stat_sum)) return ISIC_ERR_UNSUPPORTED;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.M = N * H * W;
  a.mtiles = ceil_div(a.M, HM);
  a.nslices = Cout / HN;
  int groups = cus / a.nslices;
  if (groups < 1) groups = 1;
  a.tiles_per_block = ceil_div(a.mtiles, groups);
  a.groups = ceil_div(a.mtiles, a.tiles_per_block);          // every block owns at least one tile
  a.cchunks = Cin / 64;
  a.prows = ceil_div(HM + 2 * W + 2, 8) * 8;
  a.magic_hw = ((1ULL << 40) / (unsigned long long)(H * W)) + 1;
  a.magic_w = ((1ULL << 40) / (unsigned long long)W) + 1;
  int lds = 2 * a.prows * 128 + NWST * WSTAGE + 2176 + 3072;     // statistics: 1 KB (forward) or 4 KB (STATS 2)
  const int grid = a.groups * a.nslices;
  const int lds2 = lds + WSTAGE;                             // KPB = 2: a fourth weight stage
  const bool kpb2_ok = Cin % 128 == 0 && lds2 <= 160 * 1024;
  // shipped: one barrier per two K-tiles + priority for the multiplying waves wherever the fourth weight stage fits
  // (tools/halo_ab.py, interleaved A/B at 2048 images: l2 +1 %, l3 +2..3.5 %, l4 +3..4 %); experiment 1 = the round-2/3 loop
  const bool two = kpb2_ok && experiment != 1;
  if (two) lds = lds2;
  if (experiment >= 2) {                                   // timing ablations of the shipped loop (results are garbage): ABL = experiment - 1
    if (!two || relu_mask || yraw || stat_sum || addend) return ISIC_ERR_UNSUPPORTED;
    switch (experiment - 1) {
      case 1: return launch_halo<0, false, 2, 1, 1>(a, grid, lds, stream);
      case 2: return launch_halo<0, false, 2, 1, 2>(a, grid, lds, stream);
      case 3: return launch_halo<0, false, 2, 1, 3>(a, grid, lds, stream);
      case 4: return launch_halo<0, false, 2, 1, 4>(a, grid, lds, stream);
      case 5: return launch_halo<0, false, 2, 1, 5>(a, grid, lds, stream);
      case 6: return launch_halo<0, false, 2, 1, 6>(a, grid, lds, stream);
      case 7: return launch_halo<0, false, 2, 1, 7>(a, grid, lds, stream);
      default: return ISIC_ERR_UNSUPPORTED;
    }
  }
#define HALO_LAUNCH(S, A) (two ? launch_halo<S, A, 2, 1>(a, grid, lds, stream) : launch_halo<S, A, 1, 0>(a, grid, lds, stream))
  if (relu_mask || yraw) {                                 // data gradient feeding a BatchNorm backward (STATS 2)
    if (!relu_mask || !yraw || !stat_sum || !stat_sumsq) return ISIC_ERR_BAD_ARG;
    return addend ? HALO_LAUNCH(2, true) : HALO_LAUNCH(2, false);
  }
  if (stat_sum && addend) return ISIC_ERR_UNSUPPORTED;
  if (stat_sum) return HALO_LAUNCH(1, false);
  if (addend) return HALO_LAUNCH(0, true);
  return HALO_LAUNCH(0, false);
#undef HALO_LAUNCH
}
