// Persistent implicit-GEMM convolution on bf16 MFMA (gfx950) for the SHORT-K layers of the encoder: the stride-2 3x3
// convolutions (forward and the four output-parity classes of their data gradient) and the 1x1 stride-2 downsample
// convolutions.  Same arithmetic and operand layouts as conv_igemm.hip (NHWC bf16 activations, [Cout][Kh][Kw][Cin]
// weights, one (tap, 64-channel) K-tile per step, LDS rows of 128 B with the 16-byte-chunk XOR swizzle on the DMA source).
//
// Why a second kernel: conv_igemm.hip runs ONE 256 x 128 tile per block with 144 KB of LDS, i.e. one block per CU and
// nothing to overlap a tile's pipeline fill and its epilogue with.  That is noise for the 18-72 K-tiles of the 3x3 /
// stride-1 layers, but a stride-2 3x3 layer has 9-36 K-tiles and a 1x1 downsample 1-4: fill + epilogue dominate
// (round 1: 100-470 TFLOP/s on those layers).  Here a block is persistent -- it walks a contiguous range of pixel
// tiles of one (parity class, 128-channel output slice) -- the staging waves run two K-tiles ahead ACROSS tile
// boundaries, and the epilogue is register-only (no LDS: the ring keeps streaming while the MFMA waves store).
//
// Structure (1024 threads): waves 0-7 multiply (4 x 2 waves of 64 x 64 = 4 x 4 v_mfma_f32_16x16x32_bf16 tiles, operand
// roles swapped so a lane ends with consecutive output channels of one pixel), waves 8-15 only issue LDS-DMA: per
// K-tile six 1 KB pieces each (four A pieces of eight gathered pixel rows, two B pieces of eight weight rows; an
// out-of-image tap row comes from a zero page), three 48 KB stages, one s_barrier per K-tile, hand-counted
// s_waitcnt vmcnt(6).  Epilogue as conv_halo.hip: the weight rows are permuted inside a stage so that a lane owns
// eight consecutive output channels (16-byte stores), fused residual-gradient addend (all loads before the first
// store, one rounding), BatchNorm sum / sum of squares of the rounded outputs summed per block in LDS and flushed
// once (fp64 atomics into `stat_slots` rows).
//
// The reference has no convolution kernel of its own (its encoder is an un-vendored ConvMAE run through torch,
// save_latent.py:42-60); BASELINE.json configs[1] names ResNet-18 (SURVEY.md 8d layer table).

#include "common.h"
#include "conv_args.h"

namespace {

using namespace isic_conv;

constexpr int PM = 256;                     // block tile: PM pixels x PN output channels, PN = 128 or 64 (template)
constexpr int P_A = PM * 128;               // bytes of A per K-tile
constexpr int P_NST = 3;
constexpr int p_stage(int PN) { return P_A + PN * 128; }
constexpr int p_lds(int PN) { return P_NST * p_stage(PN) + 1024 + 4096; }   // ring | DMA scratch | statistics: 8 MFMA waves x 128 floats

struct PGemmArgs {
  ConvArgsN cls;
  int nslices;          // Cout / PN
  int gfirst[5];        // class c owns blockIdx.x in [gfirst[c], gfirst[c+1]): blocks in proportion to its work
};

__device__ __attribute__((aligned(256))) unsigned char g_pg_zero_page[256];

__device__ __forceinline__ void pg_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
typedef __attribute__((ext_vector_type(2))) float pf32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 pbf16x2;
__device__ __forceinline__ unsigned pg_pack2(float lo, float hi) {
  const pf32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, pbf16x2));
}
__device__ __forceinline__ float pg_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// PN = 64 (round 2, for convolutions with 64 outputs -- the data gradient of the 64 -> 128 stride-2 layer): the MFMA
// waves stay 4 x 2 but own 64 x 32 (4 x 2 tiles), a stage holds 64 weight rows (one DMA piece per staging wave).
template <int PN, bool STATS, bool ADDEND>
__global__ __launch_bounds__(1024) void conv_pgemm_kernel(PGemmArgs pa) {
  constexpr int P_STAGE = p_stage(PN);
  constexpr int P_PER_IT = 4 + PN / 64;       // DMAs per staging wave and K-tile
  constexpr int NJ = PN / 32;                 // MFMA column tiles per wave
  constexpr int NT = PN / 64;                 // 8-channel groups per lane in the epilogue
  constexpr int WN = PN / 2;                  // columns per wave column
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  constexpr int off_scr = P_NST * P_STAGE, off_stat = off_scr + 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // block -> (parity class, output slice, contiguous range of pixel tiles).  The classes of a strided data gradient
  // visit 1, 2, 2 and 4 taps: each owns a share of the blocks in proportion to its K-tiles (host: gfirst)
  int cls = 0;
#pragma unroll
  for (int c = 1; c < 4; ++c) cls += (c < pa.cls.n && (int)blockIdx.x >= pa.gfirst[c]) ? 1 : 0;
  const ConvArgs a = pa.cls.c[cls];                        // by value: scalar registers
  const int n0 = blockIdx.y * PN;
  const int mtiles = (a.M + PM - 1) / PM;
  const int groups = pa.gfirst[cls + 1] - pa.gfirst[cls];
  const int tiles_per_block = (mtiles + groups - 1) / groups;
  const int t_begin = ((int)blockIdx.x - pa.gfirst[cls]) * tiles_per_block;
  const int ntl = min(mtiles - t_begin, tiles_per_block);
  if (ntl <= 0) return;                                    // whole block: no barrier has been reached yet
  const int KT = a.Ktiles;
  const int total_it = ntl * KT;

  typedef __attribute__((address_space(3))) float lds_float;
  lds_float* stats_lds = (lds_float*)(smem + off_stat);
  if (wave >= 8) {
    // =================================================================== staging waves
    const int sw = wave - 8;
    const int r8 = lane >> 3;
    const int gch = (lane & 7) ^ r8;                       // global 16-byte chunk this lane fetches (swizzle on the source)
    const unsigned char* zp = g_pg_zero_page + (lane & 7) * 16;
    const unsigned scr = lds0 + off_scr;
    const int ds_ = a.down_shift;
    const int dh0 = (a.oh0 * a.up + a.kh0 - a.pad) >> ds_, dw0 = (a.ow0 * a.up + a.kw0 - a.pad) >> ds_;
    const int rstep = (a.ostep * a.up) >> ds_;
    // weights: stage rows sr = 16*sw + 8*t + r8 (t = 0, 1) hold output channel
    //   wn*64 + 32*(j>>1) + 8*(rho>>2) + 4*(j&1) + (rho&3)   with wn = sr>>6, j = (sr>>4)&3, rho = sr&15
    // so that a lane's results of MFMA tiles 2t', 2t'+1 are EIGHT CONSECUTIVE channels (conv_halo.hip)
    //   PN = 64: one piece, rows sr = 8*sw + r8: wn = sr>>5, j = (sr>>4)&1, rho = sr&15 -> wn*32 + 8*(rho>>2) + 4*j + (rho&3)
    const int chan0 = PN == 128 ? (sw >> 2) * 64 + ((sw & 3) >> 1) * 32 + (r8 >> 2) * 8 + (sw & 1) * 4 + (r8 & 3)   // t = 0; t = 1: + 16
                                : (sw >> 2) * 32 + ((sw & 1) * 2 + (r8 >> 2)) * 8 + ((sw >> 1) & 1) * 4 + (r8 & 3);
    const size_t Ktot = (size_t)a.Kh * a.Kw * a.Cin;
    const unsigned short* wrow = a.w + (size_t)(n0 + chan0) * Ktot + gch * 8;

    // per-tile state of this lane's four A rows: piece i covers tile rows 8*(sw + 8*i) + r8
    int a_off[4];
    unsigned a_vh[4], a_vw[4];
    auto tile_rows = [&](int tl) {
      const int m0 = (t_begin + tl) * PM;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + 8 * (sw + 8 * i) + r8;
        a_off[i] = 0; a_vh[i] = 0u; a_vw[i] = 0u;
        if (m < a.M) {
          const int n = (int)fastdiv40((unsigned)m, a.magic_hw);
          const int r = m - n * (a.Hs * a.Ws);
          const int hs = (int)fastdiv40((unsigned)r, a.magic_w), ws = r - hs * a.Ws;
          const int hrow = hs * rstep, wrow_ = ws * rstep;
          a_off[i] = ((n * a.Hin + hrow) * a.Win + wrow_) * a.Cin + gch * 8;
          a_vh[i] = range_mask(hrow + dh0, 0, a.Hin, a.nkh);
          a_vw[i] = range_mask(wrow_ + dw0, 0, a.Win, a.nkw);
        }
      }
    };
    // the six DMAs of K-tile kt of the tile whose row state is loaded; live = false: dummy pieces (constant count)
    const unsigned short* wrow2 = a.w2 ? a.w2 + (size_t)(n0 + chan0) * a.Cin + gch * 8 : a.w;   // second source: [Cout][Cin]
    auto issue = [&](int kt, int stage, bool live) {
      const bool second = kt >= a.Kmain;                   // the joined second source (conv_args.h): tap (0, 0), own weights
      const int tap = second ? 0 : kt / a.ctiles, c0 = ((second ? kt - a.Kmain : kt) - tap * a.ctiles) * 64;
      const int ti = tap / a.nkw, tj = tap - ti * a.nkw;
      const int toff = ((dh0 + ti) * a.Win + (dw0 + tj)) * a.Cin + c0;
      const int koff = second ? c0 : ((a.kh0 + a.kstep * ti) * a.Kw + (a.kw0 + a.kstep * tj)) * a.Cin + c0;
      const unsigned short* inp = second ? a.in2 : a.in;
      const unsigned short* wp = second ? wrow2 : wrow;
      const size_t wstep = second ? (size_t)16 * a.Cin : (size_t)16 * Ktot;
      const unsigned sbase = lds0 + stage * P_STAGE;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = live && (((a_vh[i] >> ti) & (a_vw[i] >> tj)) & 1u);
        const void* src = ok ? (const void*)(inp + (a_off[i] + toff)) : (const void*)zp;
        pg_glds16(src, live ? sbase + (unsigned)((sw + 8 * i) * 1024) : scr);
      }
      if (PN == 128) {
        pg_glds16(live ? (const void*)(wp + koff) : (const void*)zp, live ? sbase + P_A + sw * 2048 : scr);
        pg_glds16(live ? (const void*)(wp + koff + wstep) : (const void*)zp, live ? sbase + P_A + sw * 2048 + 1024 : scr);
      } else {
        pg_glds16(live ? (const void*)(wp + koff) : (const void*)zp, live ? sbase + P_A + sw * 1024 : scr);
      }
    };

    if (KT > 0) {
      // the issue cursor (itile, ikt) runs two K-tiles ahead of the multiply; the row state follows the cursor's tile
      int itile = 0, ikt = 0;
      tile_rows(0);
      auto advance = [&]() {
        if (++ikt == KT) { ikt = 0; ++itile; if (itile < ntl) tile_rows(itile); }
      };
      issue(0, 0, true); advance();
      issue(ikt, 1, total_it > 1); advance();
      for (int it = 0; it < total_it; ++it) {
        // K-tile it (issued two iterations ago, or in the prologue) has landed when only the previous iteration's six
        // DMAs are still in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_PER_IT) : "memory");
        __builtin_amdgcn_s_barrier();          // ... for every staging wave; the MFMA waves are done with K-tile it-1
        issue(ikt, (it + 2) % P_NST, it + 2 < total_it);
        advance();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA may outlive the block's LDS allocation
  } else {
    // ===================================================================== MFMA waves
    if (STATS && tid < 512) { stats_lds[tid] = 0.f; stats_lds[tid + 512] = 0.f; }      // 8 waves x 128 floats
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    const int fr = lane & 15, fg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // fragment byte offsets inside a stage (rows 16 apart share the swizzle key row & 7: tile i / j = + 2048 i / j)
    const unsigned aoff0 = (unsigned)((wm * 64 + fr) * 128 + ((fg ^ (fr & 7)) << 4));
    const unsigned boff0 = (unsigned)(P_A + (wn * WN + fr) * 128 + ((fg ^ (fr & 7)) << 4));
    const bool dense = (a.ostep == 1 && a.Hs == a.Hout && a.Ws == a.Wout);

    int it = 0;
    for (int tl = 0; tl < ntl; ++tl) {
      const int m0 = (t_begin + tl) * PM;
      f32x4 acc[4][NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
      for (int kt = 0; kt < KT; ++kt, ++it) {
        __builtin_amdgcn_s_barrier();
        const unsigned char* st = smem + (it % P_NST) * P_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 af[4], bfr[NJ];
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + ((aoff0 ^ (unsigned)(ks << 6)) + i * 2048));
#pragma unroll
          for (int j = 0; j < NJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(st + ((boff0 ^ (unsigned)(ks << 6)) + j * 2048));
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
      }

      // ---- register-only epilogue: lane (fg, fr) holds, for MFMA tiles (i, 2t) and (i, 2t+1), the eight consecutive
      //      output channels n0 + wn*64 + 32t + 8fg + {0..7} of sub-grid pixel m0 + wm*64 + i*16 + fr
      const unsigned chan = (unsigned)(n0 + wn * WN + fg * 8);
      unsigned off[4];
      bool valid[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        valid[i] = m < a.M;
        unsigned pix = (unsigned)(valid[i] ? m : 0);
        if (!dense && valid[i]) {
          const int n = (int)fastdiv40((unsigned)m, a.magic_hw);
          const int r = m - n * (a.Hs * a.Ws);
          const int hs = (int)fastdiv40((unsigned)r, a.magic_w), ws = r - hs * a.Ws;
          pix = (unsigned)((n * a.Hout + (a.oh0 + a.ostep * hs)) * a.Wout + (a.ow0 + a.ostep * ws));
        }
        off[i] = pix * (unsigned)a.Cout + chan;            // N*Hout*Wout*Cout < 2^31 (host check)
      }
      u32x4 ad[4][NT];
      if (ADDEND) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t) ad[i][t] = *reinterpret_cast<const u32x4*>(a.addend + off[i] + t * 32);
      }
      // DETERMINISTIC statistics (round 3): a wave adds to its OWN LDS slots in tile order (no LDS atomics in arrival
      // order); the flush adds the four wm waves of a channel column in a fixed order, one fp64 slot row per block
      lds_float* sp = stats_lds + wave * 128 + lane;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float s8[8], q8[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { s8[c] = 0.f; q8[c] = 0.f; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u32x4 v;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x4 c = acc[i][2 * t + h];
            if (ADDEND) {
              const unsigned lo = ad[i][t][2 * h], hi = ad[i][t][2 * h + 1];
              c[0] += __uint_as_float(lo << 16);
              c[1] += __uint_as_float(lo & 0xFFFF0000u);
              c[2] += __uint_as_float(hi << 16);
              c[3] += __uint_as_float(hi & 0xFFFF0000u);
            }
            const unsigned w0 = pg_pack2(c[0], c[1]), w1 = pg_pack2(c[2], c[3]);
            v[2 * h] = w0;
            v[2 * h + 1] = w1;
            if (STATS && valid[i]) {                       // statistics of the ROUNDED outputs
              const float r0 = __uint_as_float(w0 << 16), r1 = __uint_as_float(w0 & 0xFFFF0000u);
              const float r2 = __uint_as_float(w1 << 16), r3 = __uint_as_float(w1 & 0xFFFF0000u);
              s8[4 * h + 0] += r0; q8[4 * h + 0] += r0 * r0;
              s8[4 * h + 1] += r1; q8[4 * h + 1] += r1 * r1;
              s8[4 * h + 2] += r2; q8[4 * h + 2] += r2 * r2;
              s8[4 * h + 3] += r3; q8[4 * h + 3] += r3 * r3;
            }
          }
          // ordinary stores (late round 4): a wave writes a pixel's 64-byte segment per instruction, its partner -- the next t or the
          // other channel-half wave -- the rest of the 128-byte line; cached, the L2 joins them (non-temporal, each segment went
          // to memory by itself: layer2.0 pair gradient 1.299 -> 1.267 ms at 4096 images)
          if (valid[i]) *reinterpret_cast<u32x4*>(a.out + off[i] + t * 32) = v;
        }
        if (STATS) {
          // every lane of a DPP row (16 pixels) ends with the row totals of the 16 values (8 sums, 8 sums of squares);
          // lane fr keeps value #fr (fr < 8: sum of channel fg*8 + fr; else sum of squares of channel fg*8 + fr - 8)
          float mine = 0.f;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float sv = pg_row16_sum(s8[c]), qv = pg_row16_sum(q8[c]);
            mine = fr == c ? sv : mine;
            mine = fr == 8 + c ? qv : mine;
          }
          sp[t * 64] += mine;
        }
      }
    }
  }   // MFMA waves

  if (STATS) {
    lds_barrier();                                         // all sixteen waves: every tile's partial sums are in LDS
    if (tid < 2 * PN) {
      const size_t slot = (size_t)((blockIdx.x + blockIdx.y) % a.stat_slots) * a.Cout + n0 + (tid & (PN - 1));
      // channel c of the slice = wn*WN + t*32 + fg*8 + e lives in lane fg*16 + e (+ 8 for the sum of squares) of the
      // slots of waves wm*2 + wn, wm = 0..3
      const int c = tid & (PN - 1), wn_ = c / WN, t = (c % WN) >> 5, l = ((c >> 3) & 3) * 16 + (c & 7) + (tid < PN ? 0 : 8);
      float v = 0.f;
#pragma unroll
      for (int wm_ = 0; wm_ < 4; ++wm_) v += stats_lds[(wm_ * 2 + wn_) * 128 + t * 64 + l];
      // (with stat_slots >= gridDim.x the slot row is this block's own: 0 + v is exact, bit-reproducible)
      atomicAdd((tid < PN ? a.stat_sum : a.stat_sumsq) + slot, (double)v);
    }
  }
}

template <int PN, bool STATS, bool ADDEND>
int launch_pgemm(const PGemmArgs& pa, dim3 grid, hipStream_t stream) {
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (isic_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pgemm_kernel<PN, STATS, ADDEND>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, p_lds(PN));
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL((conv_pgemm_kernel<PN, STATS, ADDEND>), grid, dim3(1024), p_lds(PN), stream, pa);
  return ISIC_OK;
}

}  // namespace

// Called by isic_conv2d_igemm_bf16 (conv_igemm.hip) with the parity classes it has set up; Cout % 64 == 0
// (128-channel slices when Cout % 128 == 0, else 64-channel slices).
int isic_conv_pgemm_launch(const isic_conv::ConvArgsN& classes, hipStream_t stream) {
  const int cus = isic_cu_count();
  const ConvArgs& a0 = classes.c[0];
  if (a0.Cout % 64 != 0 || classes.n < 1) return ISIC_ERR_UNSUPPORTED;
  const int PN = a0.Cout % 128 == 0 ? 128 : 64;
  if (a0.stat_sum && a0.addend) return ISIC_ERR_UNSUPPORTED;
  PGemmArgs pa;
  pa.cls = classes;
  pa.nslices = a0.Cout / PN;
  // blocks per class in proportion to its work: tiles x (K-tiles + a tile's fixed cost, ~6 K-tiles of epilogue)
  int budget = cus / pa.nslices;
  if (budget < classes.n) budget = classes.n;
  double cost[4], total = 0.0;
  int mt[4];
  for (int i = 0; i < classes.n; ++i) {
    mt[i] = (classes.c[i].M + PM - 1) / PM;
    cost[i] = (double)mt[i] * (classes.c[i].Ktiles + (a0.stat_sum ? 12.0 : 6.0));
    total += cost[i];
  }
  int g[4], used = 0;
  for (int i = 0; i < classes.n; ++i) {
    g[i] = (int)(budget * cost[i] / total);
    if (g[i] < 1) g[i] = 1;
    if (g[i] > mt[i]) g[i] = mt[i];
    used += g[i];
  }
  // hand the blocks lost to rounding to the classes with the most work per block
  for (int left = budget - used; left > 0; --left) {
    int best = -1;
    double load = 0.0;
    for (int i = 0; i < classes.n; ++i)
      if (g[i] < mt[i] && cost[i] / g[i] > load) { load = cost[i] / g[i]; best = i; }
    if (best < 0) break;
    ++g[best];
  }
  pa.gfirst[0] = 0;
  for (int i = 0; i < 4; ++i) pa.gfirst[i + 1] = pa.gfirst[i] + (i < classes.n ? g[i] : 0);
  const dim3 grid(pa.gfirst[classes.n], pa.nslices);
  if (PN == 128) {
    if (a0.stat_sum) return launch_pgemm<128, true, false>(pa, grid, stream);
    if (a0.addend) return launch_pgemm<128, false, true>(pa, grid, stream);
    return launch_pgemm<128, false, false>(pa, grid, stream);
  }
  if (a0.stat_sum) return launch_pgemm<64, true, false>(pa, grid, stream);
  if (a0.addend) return launch_pgemm<64, false, true>(pa, grid, stream);
  return launch_pgemm<64, false, false>(pa, grid, stream);
}
