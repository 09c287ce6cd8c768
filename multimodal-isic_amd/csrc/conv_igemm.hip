// Implicit-GEMM convolution on bf16 MFMA (gfx950), NHWC activations, [Cout][Kh][Kw][Cin] weights.
//
//   out[m, co] = sum_{kh,kw,ci} in[pix(m,kh,kw), ci] * w[co, kh, kw, ci]  (+ addend[m, co])   m = (n, ho, wo)
//
// GEMM view: M = N*Hout*Wout (pixels), N = Cout, K = Kh*Kw*Cin with BOTH operands K-contiguous
// (NHWC rows are channel-contiguous, weights are stored K-major), which is exactly the operand
// shape v_mfma_f32_16x16x32_bf16 wants: 8 consecutive k per lane = one 16-byte load.
//
// The same kernel is the forward pass (up = stride, down = 1) and the data-gradient pass
// (up = 1, down = stride, flipped/transposed weights): the source pixel of tap (kh,kw) for output
// (ho,wo) is ((ho*up + kh - pad)/down, (wo*up + kw - pad)/down).  A stride-2 data gradient is
// split by output-pixel parity into 4 launches, each visiting only the taps whose source pixel is
// integral for that parity class (9 taps in total instead of 36 for a 3x3 kernel).
//
// Fused epilogues (all optional):
//   * addend: out = conv + addend, added in fp32 BEFORE the single rounding to bf16 (the residual
//     gradient join of a ResNet block: d_x = dgrad(conv1) + d_residual);
//   * BatchNorm statistics: per-channel sum / sum of squares of the bf16-rounded outputs of the
//     tile, accumulated with fp64 atomics into one of `stat_slots` partial rows (slot =
//     blockIdx.x % slots) -- replaces a full read pass over the conv output.
//
// The reference has no convolution kernel of its own (its encoder is an un-vendored ConvMAE run
// through torch, save_latent.py:42-60); BASELINE.json configs[1] names ResNet-18, whose layer table
// is SURVEY.md 8d.
//
// Structure: 256 threads = 4 waves; block tile BM x BN (256x64 as 4x1 waves for Cout = 64,
// 128x128 as 2x2 waves otherwise), each wave a 64x64 sub-tile = 4x4 MFMA tiles (64 accumulator
// VGPRs); BK = 64 = one (kh,kw) tap x 64 channels.  Staging is LDS-DMA (global_load_lds, 16 B per
// lane, no staging VGPRs, no ds_write) into two LDS stages: tile k+1 streams in while tile k is
// multiplied, one barrier per K-tile, two blocks per CU.  LDS rows are 128 B with a 16-byte-chunk XOR
// swizzle applied on the SOURCE side (the DMA writes lane-linearly) so the ds_read_b128 fragment
// reads are conflict-free; out-of-image taps are fetched from a zero page.  The epilogue transposes
// through LDS so every global store is a full 16-byte-per-lane row segment.  (Register-staged and
// 3-stage variants are kept behind ISIC_CONV_MODE for A/B timing; measured slower.)
#include <stdlib.h>


#include "common.h"
#include "conv_args.h"

namespace {

using namespace isic_conv;

constexpr int BK = 64;  // bf16 elements per K-tile (128 bytes per LDS row)

__device__ __forceinline__ float bfbits(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// zero page for LDS-DMA staging: an out-of-image tap row is fetched from here instead of being zero-filled
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

// MODE 0: one LDS stage, register staging      MODE 1: two LDS stages, register staging
// MODE 2: two LDS stages, LDS-DMA staging       MODE 3: three LDS stages, LDS-DMA, counted vmcnt
// MODE 4: as MODE 2 with dedicated STAGING WAVES behind the MFMA waves: they only issue the LDS-DMA (an LDS-DMA
//         instruction holds its wave for the order of 100 cycles at issue), the MFMA waves only read LDS and multiply
// MODE 5: staging waves and THREE stages (the DMA of K-tile kt+2 is issued while kt is multiplied: with two stages
//         the DMA latency of every K-tile is exposed behind the barrier); used with the 256 x 128 tile, 8 + 8 waves
#ifdef IGEMM_STAMPS   // tests/probes/probe_igemm_stamps.hip: per K-tile clocks of one MFMA wave (slots 0-2) and one staging wave (3-6)
__device__ unsigned long long g_igemm_stamps[256 * 40 * 8];
#define IG_STAMP(kt, k, who) do { if (threadIdx.x == (who) && blockIdx.x < 256 && blockIdx.y == 0 && (kt) < 40) g_igemm_stamps[(blockIdx.x * 40 + (kt)) * 8 + (k)] = clock64(); } while (0)
#else
#define IG_STAMP(kt, k, who) do { } while (0)
#endif

template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE>
__global__ __launch_bounds__((MODE >= 4 ? 2 : 1) * WAVES_M * WAVES_N * 64, MODE >= 4 ? 4 : 1) void conv_igemm_kernel(ConvArgsN classes) {
  const ConvArgs a = classes.c[blockIdx.z];           // by value: loaded into scalar registers once
  if ((int)(blockIdx.x * BM) >= a.M) return;            // this class has fewer row tiles than the largest one (whole block)
  constexpr bool DB = (MODE == 1);
  constexpr bool GLDS = (MODE >= 2);
  constexpr int NSTAGE = (MODE == 3 || MODE == 5) ? 3 : (MODE == 0 ? 1 : 2);
  constexpr int MT = WAVES_M * WAVES_N * 64;          // MFMA threads; MODE >= 4: as many staging threads behind them
  static_assert(BM == WAVES_M * 64 && BN == WAVES_N * 64, "64x64 per wave");
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int CPAD = BN + 8;  // epilogue row stride (elements)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NSTAGE*STAGE, >= C tile

  const bool stager = MODE >= 4 && threadIdx.x >= MT;
  const int tid = stager ? threadIdx.x - MT : threadIdx.x;   // index inside the role
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  // ---- staging assignment: 16-byte chunk `sc` of rows `sr + 32*i`.
  // Everything that depends on the ROW is computed once, before the K loop: the source pixel of tap
  // (ti,tj) is (hrow + dh(ti), wrow + dw(tj)) with a per-tap uniform (dh, dw), so a row carries its
  // element offset at (dh, dw) = (0, 0) plus one validity bit per tap row / tap column; inside the K loop a
  // row costs two bit tests, one add and one select.
  constexpr int RPP = MT / 8;                          // rows staged per pass: 8 threads per 128-byte row
  constexpr int AROWS = BM / RPP, BROWS = BN / RPP;
  const int sr = tid >> 3;
  const int sc = GLDS ? ((tid & 7) ^ (sr & 7)) : (tid & 7);   // global 16-byte chunk this thread fetches
  const int ds_ = a.down_shift;
  // per-tap uniform offsets: dh(ti) = (oh0*up + kh - pad) >> ds  (exact for the taps of this parity class)
  const int dh0 = (a.oh0 * a.up + a.kh0 - a.pad) >> ds_, dw0 = (a.ow0 * a.up + a.kw0 - a.pad) >> ds_;
  constexpr int dstep = 1;            // kstep >> down_shift: 1 in both modes (kstep = down)
  const int rstep = (a.ostep * a.up) >> ds_;   // source rows per sub-grid row: stride (fwd) or 1 (dgrad class)
  int a_off[AROWS];          // element offset of (n, hrow, wrow, 0); meaningless when no tap is valid
  unsigned a_vh[AROWS], a_vw[AROWS];   // bit ti: 0 <= hrow + dh0 + dstep*ti < Hin ; bit tj likewise for columns
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + sr + RPP * i;
    a_off[i] = 0; a_vh[i] = 0u; a_vw[i] = 0u;
    if (m < a.M) {
      const int n = (int)fastdiv40((unsigned)m, a.magic_hw);
      const int r = m - n * (a.Hs * a.Ws);
      const int hs = (int)fastdiv40((unsigned)r, a.magic_w), ws = r - hs * a.Ws;
      const int hrow = hs * rstep, wrow = ws * rstep;
      a_off[i] = ((n * a.Hin + hrow) * a.Win + wrow) * a.Cin + sc * 8;
      a_vh[i] = range_mask(hrow + dh0, 0, a.Hin, a.nkh);     // dstep == 1
      a_vw[i] = range_mask(wrow + dw0, 0, a.Win, a.nkw);
    }
  }
  const size_t Ktot = (size_t)a.Kh * a.Kw * a.Cin;
  const unsigned short* wrow_ptr[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) wrow_ptr[i] = a.w + (size_t)(n0 + sr + RPP * i) * Ktot + sc * 8;

  u32x4 ra0[AROWS], rb0[BROWS];
  auto gload = [&](int kt, u32x4 (&ra)[AROWS], u32x4 (&rb)[BROWS]) {
    const int tap = kt / a.ctiles, c0 = (kt - tap * a.ctiles) * BK;
    const int ti = tap / a.nkw, tj = tap - ti * a.nkw;
    const int toff = ((dh0 + dstep * ti) * a.Win + (dw0 + dstep * tj)) * a.Cin + c0;          // uniform
    const int koff = ((a.kh0 + a.kstep * ti) * a.Kw + (a.kw0 + a.kstep * tj)) * a.Cin + c0;     // uniform
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (((a_vh[i] >> ti) & (a_vw[i] >> tj)) & 1u) v = *reinterpret_cast<const u32x4*>(a.in + (a_off[i] + toff));
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) rb[i] = *reinterpret_cast<const u32x4*>(wrow_ptr[i] + koff);
  };
  auto lstore = [&](unsigned char* As, unsigned char* Bs, const u32x4 (&ra)[AROWS], const u32x4 (&rb)[BROWS]) {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int r = sr + RPP * i;
      *reinterpret_cast<u32x4*>(As + r * 128 + ((sc ^ (r & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const int r = sr + RPP * i;
      *reinterpret_cast<u32x4*>(Bs + r * 128 + ((sc ^ (r & 7)) << 4)) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](const unsigned char* As, const unsigned char* Bs) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra_ = wm * 64 + i * 16 + fr;
        af[i] = *reinterpret_cast<const bf16x8*>(As + ra_ * 128 + (((ks * 4 + fg) ^ (ra_ & 7)) << 4));
        const int rb_ = wn * 64 + i * 16 + fr;
        bfr[i] = *reinterpret_cast<const bf16x8*>(Bs + rb_ * 128 + (((ks * 4 + fg) ^ (rb_ & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  if (GLDS) {
    // LDS-DMA staging (global_load_lds, 16 B per lane): a wave-instruction fills 8 consecutive 128-byte rows
    // lane-linearly, so lane (r8 = lane>>3, p = lane&7) supplies global chunk p ^ r8 of row 8*g + r8 -- the same
    // swizzled image the register path builds.  No staging VGPRs, no ds_write; NSTAGE-1 tiles in flight.
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    const unsigned char* zp = g_zero_page + (tid & 7) * 16;
    auto issue = [&](int kt, unsigned char* stage) {
      const bool second = kt >= a.Kmain;                   // the joined second source (conv_args.h): tap (0, 0), own weights
      const int tap = second ? 0 : kt / a.ctiles, c0 = ((second ? kt - a.Kmain : kt) - tap * a.ctiles) * BK;
      const int ti = tap / a.nkw, tj = tap - ti * a.nkw;
      const int toff = ((dh0 + dstep * ti) * a.Win + (dw0 + dstep * tj)) * a.Cin + c0;          // uniform
      const int koff = second ? c0 : ((a.kh0 + a.kstep * ti) * a.Kw + (a.kw0 + a.kstep * tj)) * a.Cin + c0;     // uniform
      const unsigned short* inp = second ? a.in2 : a.in;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const bool ok = ((a_vh[i] >> ti) & (a_vw[i] >> tj)) & 1u;
        const void* src = ok ? (const void*)(inp + (a_off[i] + toff)) : (const void*)zp;
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(stage + (wave * 8 + RPP * i) * 128), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const unsigned short* wp = second ? a.w2 + (size_t)(n0 + sr + RPP * i) * a.Cin + sc * 8 : wrow_ptr[i];
        __builtin_amdgcn_global_load_lds((gbl_ptr)(wp + koff), (lds_ptr)(stage + A_BYTES + (wave * 8 + RPP * i) * 128), 16, 0, 0);
      }
    };
    constexpr int LPT = AROWS + BROWS;        // LDS-DMA instructions per wave per K-tile
    if (MODE >= 4) {
      // stage of K-tile kt: kt % NSTAGE; the staging waves run NSTAGE-1 tiles ahead of the MFMA waves
      if (stager) {
#pragma unroll
        for (int s_ = 0; s_ < NSTAGE - 1; ++s_)
          if (s_ < a.Ktiles) issue(s_, smem + s_ * STAGE);
        for (int kt = 0; kt < a.Ktiles; ++kt) {
          IG_STAMP(kt, 3, MT);
          // tile kt landed (this wave's rows) when only the younger tiles' DMAs are in flight
          if (NSTAGE == 3 && kt + 1 < a.Ktiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          IG_STAMP(kt, 4, MT);
          __builtin_amdgcn_s_barrier();                        // ... every row; and the MFMA waves are done with tile kt-1
          IG_STAMP(kt, 5, MT);
          if (kt + NSTAGE - 1 < a.Ktiles) issue(kt + NSTAGE - 1, smem + ((kt + NSTAGE - 1) % NSTAGE) * STAGE);
          IG_STAMP(kt, 6, MT);
        }
        __builtin_amdgcn_s_barrier();                          // matches the MFMA waves' barrier behind the K loop
        return;                                                // the epilogue belongs to the MFMA waves
      }
      for (int kt = 0; kt < a.Ktiles; ++kt) {
        IG_STAMP(kt, 0, 0);
        __builtin_amdgcn_s_barrier();
        IG_STAMP(kt, 1, 0);
        unsigned char* As = smem + (kt % NSTAGE) * STAGE;
        compute(As, As + A_BYTES);
        IG_STAMP(kt, 2, 0);
      }
      __builtin_amdgcn_s_barrier();                            // all MFMA waves are done reading the stages
    } else {
#pragma unroll
    for (int s_ = 0; s_ < NSTAGE - 1; ++s_)
      if (s_ < a.Ktiles) issue(s_, smem + s_ * STAGE);
    for (int kt = 0; kt < a.Ktiles; ++kt) {
      // tile kt landed when at most the (NSTAGE-2) younger tiles' loads are still outstanding
      if (NSTAGE == 3 && kt + 1 < a.Ktiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();           // everyone's part landed; everyone is done with stage (kt-1) % NSTAGE
      if (kt + NSTAGE - 1 < a.Ktiles) issue(kt + NSTAGE - 1, smem + ((kt + NSTAGE - 1) % NSTAGE) * STAGE);
      unsigned char* As = smem + (kt % NSTAGE) * STAGE;
      compute(As, As + A_BYTES);
    }
    __syncthreads();
    }
  } else if (DB) {
    // variant 1: two LDS stages, one register set, one barrier per K-tile
    if (a.Ktiles > 0) {   // a parity class of a strided data gradient can have no tap at all
      gload(0, ra0, rb0);
      lstore(smem, smem + A_BYTES, ra0, rb0);
    }
    __syncthreads();
    for (int kt = 0; kt < a.Ktiles; ++kt) {
      unsigned char* As = smem + (kt & 1) * STAGE;
      const bool more = kt + 1 < a.Ktiles;
      if (more) gload(kt + 1, ra0, rb0);
      compute(As, As + A_BYTES);
      if (more) {
        unsigned char* An = smem + ((kt + 1) & 1) * STAGE;
        lstore(An, An + A_BYTES, ra0, rb0);
      }
      __syncthreads();
    }
  } else {
    // variant 0: one LDS stage, next tile's global loads in flight during the MFMAs
    if (a.Ktiles > 0) gload(0, ra0, rb0);
    for (int kt = 0; kt < a.Ktiles; ++kt) {
      __syncthreads();                      // previous tile fully consumed
      lstore(smem, smem + A_BYTES, ra0, rb0);
      __syncthreads();
      if (kt + 1 < a.Ktiles) gload(kt + 1, ra0, rb0);
      compute(smem, smem + A_BYTES);
    }
    __syncthreads();
  }

  // ---- epilogue.  The MFMAs were issued as D[co][pixel] = W * X^T, so lane (fg, fr) holds, for MFMA tile
  // (i, j), the FOUR CONSECUTIVE output channels co = j*16 + fg*4 + {0..3} of pixel i*16 + fr: one packed
  // 8-byte LDS store per tile instead of four 2-byte ones.
  unsigned short* Cs = reinterpret_cast<unsigned short*>(smem);
  constexpr int CHUNKS = BN / 8;  // 16-byte chunks per output row
  auto out_index = [&](int m) -> size_t {   // flat pixel index of sub-grid row m in the full output
    const int n = (int)fastdiv40((unsigned)m, a.magic_hw);
    const int r = m - n * (a.Hs * a.Ws);
    const int hs = (int)fastdiv40((unsigned)r, a.magic_w), ws = r - hs * a.Ws;
    return ((size_t)n * a.Hout + (a.oh0 + a.ostep * hs)) * a.Wout + (a.ow0 + a.ostep * ws);
  };
  const bool dense = (a.ostep == 1 && a.Hs == a.Hout && a.Ws == a.Wout);
  const int crow0 = (wm * 64 + fr) * CPAD + wn * 64 + fg * 4;     // element index of tile (0,0)'s 4 values
  if (a.addend) {
    // stage the addend tile (coalesced), add in fp32 in the accumulator domain: ONE rounding
    for (int idx = tid; idx < BM * CHUNKS; idx += MT) {
      const int row = idx / CHUNKS, ch = idx - row * CHUNKS;
      const int m = m0 + row;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (m < a.M) v = *reinterpret_cast<const u32x4*>(a.addend + (dense ? (size_t)m : out_index(m)) * a.Cout + n0 + ch * 8);
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
      u32x2 v;
      v[0] = (unsigned)f32_to_bf16_bits(acc[i][j][0]) | ((unsigned)f32_to_bf16_bits(acc[i][j][1]) << 16);
      v[1] = (unsigned)f32_to_bf16_bits(acc[i][j][2]) | ((unsigned)f32_to_bf16_bits(acc[i][j][3]) << 16);
      *reinterpret_cast<u32x2*>(Cs + crow0 + i * 16 * CPAD + j * 16) = v;
    }
  lds_barrier();
  if (a.stat_sum) {
    // per-channel sum / sum of squares of the ROUNDED outputs of this tile (rows >= M are exact zeros);
    // done BEFORE the output stores so that no barrier has to wait for stores in flight
    constexpr int PARTS = MT / BN;            // threads per column
    constexpr int RPP = BM / PARTS;           // rows per thread
    const int col = tid % BN, part = tid / BN;
    float s = 0.f, q = 0.f;
#pragma unroll 16
    for (int r = part * RPP; r < (part + 1) * RPP; ++r) {
      const float v = bfbits(Cs[r * CPAD + col]);
      s += v; q += v * v;
    }
    float* red = reinterpret_cast<float*>(smem + BM * CPAD * 2);   // behind the C tile: 2*MT floats
    red[tid] = s; red[MT + tid] = q;
    lds_barrier();
    if (tid < BN) {
      double ds = 0.0, dq = 0.0;
#pragma unroll
      for (int p = 0; p < PARTS; ++p) { ds += (double)red[p * BN + tid]; dq += (double)red[MT + p * BN + tid]; }
      const size_t slot = (size_t)(blockIdx.x % a.stat_slots) * a.Cout + n0 + tid;
      atomicAdd(a.stat_sum + slot, ds);
      atomicAdd(a.stat_sumsq + slot, dq);
    }
  }
  for (int idx = tid; idx < BM * CHUNKS; idx += MT) {
    const int row = idx / CHUNKS, ch = idx - row * CHUNKS;
    const int m = m0 + row;
    if (m < a.M) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(Cs + row * CPAD + ch * 8);
      const size_t pix = dense ? (size_t)m : out_index(m);
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(a.out + pix * a.Cout + n0 + ch * 8));   // streaming: not re-read here
    }
  }
}

// master fp32 weights in [O][Kh][Kw][I] memory order (torch channels_last of an OIHW tensor) ->
// bf16 forward copy (same order) and bf16 dgrad copy [I][Kh][Kw][O] with both taps flipped.
__global__ void weight_prep_kernel(const float* __restrict__ w, unsigned short* __restrict__ w_fwd,
                                   unsigned short* __restrict__ w_dgrad, int O, int I, int Kh, int Kw) {
  const int64_t n = (int64_t)O * Kh * Kw * I;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % I);
    int64_t t = idx / I;
    const int kw = (int)(t % Kw); t /= Kw;
    const int kh = (int)(t % Kh);
    const int o = (int)(t / Kh);
    const unsigned short b = f32_to_bf16_bits(w[idx]);
    if (w_fwd) w_fwd[idx] = b;
    if (w_dgrad) w_dgrad[(((int64_t)i * Kh + (Kh - 1 - kh)) * Kw + (Kw - 1 - kw)) * O + o] = b;
  }
}

template <int BM, int BN, int WM, int WN, int MODE>
int launch_conv(const ConvArgsN& a, hipStream_t s) {
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int MT = WM * WN * 64;
  constexpr int CBYTES = BM * (BN + 8) * 2 + 2 * MT * 4;
  constexpr int MAIN = ((MODE == 3 || MODE == 5) ? 3 : (MODE == 0 ? 1 : 2)) * STAGE;
  constexpr int THREADS = (MODE >= 4 ? 2 : 1) * MT;
  constexpr int LDS = MAIN > CBYTES ? MAIN : CBYTES;
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (isic_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  int maxM = 0;
  for (int i = 0; i < a.n; ++i) maxM = a.c[i].M > maxM ? a.c[i].M : maxM;
  dim3 grid(ceil_div(maxM, BM), a.c[0].Cout / BN, a.n);
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, MODE>), grid, dim3(THREADS), LDS, s, a);
  return ISIC_OK;
}

template <int BM, int BN, int WM, int WN>
int launch_conv_mode(int mode, const ConvArgsN& a, hipStream_t s) {
  switch (mode) {
    case 1: return launch_conv<BM, BN, WM, WN, 1>(a, s);
    case 2: return launch_conv<BM, BN, WM, WN, 2>(a, s);
    case 3: return launch_conv<BM, BN, WM, WN, 3>(a, s);
    case 4: return launch_conv<BM, BN, WM, WN, 4>(a, s);
    default: return launch_conv<BM, BN, WM, WN, 0>(a, s);
  }
}

constexpr int kDefaultConvMode = 5;   // staging waves, three stages, 256 x 128 tile: fastest measured (tools/kernel_bench.py)
constexpr int kDefaultConvC64 = 2;    // 64 -> 64 3x3 layers: persistent halo kernel (conv_c64.hip)

}  // namespace

int isic_conv3x3_c64_launch(int variant, const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int H, int W,
                            const uint16_t* addend, const uint8_t* addend_mask, double* stat_sum, double* stat_sumsq,
                            int stat_slots, int experiment, hipStream_t stream);
bool isic_conv_halo_supported(int N, int H, int W, int Cin, int Cout);
int isic_conv_halo_launch(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int H, int W, int Cin, int Cout,
                          const uint16_t* addend, const uint8_t* addend_mask, double* stat_sum, double* stat_sumsq,
                          int stat_slots, const uint8_t* relu_mask, const uint16_t* yraw, int experiment, hipStream_t stream);
int isic_conv_pgemm_launch(const isic_conv::ConvArgsN& classes, hipStream_t stream);

namespace {

// `variant` (include/isic_hip_test.h): 0 = the shipped dispatch.  Otherwise, decimal digits
//   units: staging MODE of the generic kernel + 1 (0 = default), tens: 64 -> 64 kernels (0 default, 1 generic kernel,
//   2 tile per block, 3 persistent), hundreds: pixels-staged-once kernel of conv_halo.hip (0 = where profitable,
//   1 = never, 2 = wherever it is supported), thousands: persistent short-K kernel of conv_pgemm.hip (0 = where
//   profitable: strided and 1x1 layers, 1 = never, 2 = wherever it is supported).  No global state: the choice travels
//   with the call.
struct ConvVariant {
  int mode, c64, halo, pgemm, exp;
};
inline ConvVariant decode_variant(int v) {
  ConvVariant r;
  const int m = v % 10, c = (v / 10) % 10;
  r.mode = m == 0 ? kDefaultConvMode : m - 1;
  r.c64 = c == 0 ? kDefaultConvC64 : c - 1;
  r.halo = (v / 100) % 10;
  r.pgemm = (v / 1000) % 10;
  r.exp = (v / 10000) % 10;          // ten-thousands: a kernel-internal A/B experiment (0 = shipped code)
  return r;
}

int conv2d_dispatch(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int Hin, int Win, int Cin,
                    int Hout, int Wout, int Cout, int Kh, int Kw, int up, int down, int pad,
                    const uint16_t* addend, double* stat_sum, double* stat_sumsq, int stat_slots, int variant,
                    void* stream, const uint8_t* addend_mask = nullptr, const uint16_t* in2 = nullptr,
                    const uint16_t* w2 = nullptr);

}  // namespace

namespace {

int conv2d_dispatch(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int Hin, int Win, int Cin,
                    int Hout, int Wout, int Cout, int Kh, int Kw, int up, int down, int pad,
                    const uint16_t* addend, double* stat_sum, double* stat_sumsq, int stat_slots, int variant,
                    void* stream, const uint8_t* addend_mask, const uint16_t* in2, const uint16_t* w2) {
  ISIC_CHECK_ARG(in && w && out);
  ISIC_CHECK_ARG(!addend_mask || addend);
  ISIC_CHECK_ARG((in2 == nullptr) == (w2 == nullptr));
  // a second source exists for the stride-2 3x3 data gradient only (pad' = 1: its even-pixel class has the single tap (1, 1))
  if (in2 && !(Kh == 3 && Kw == 3 && up == 1 && down == 2 && pad == 1 && !addend && !stat_sum)) return ISIC_ERR_UNSUPPORTED;
  ISIC_CHECK_ARG(variant >= 0 && variant < 100000);
  const ConvVariant cv = decode_variant(variant);
  ISIC_CHECK_ARG(cv.mode >= 0 && cv.mode <= 5 && cv.c64 >= 0 && cv.c64 <= 2 && cv.halo >= 0 && cv.halo <= 2 &&
                 cv.pgemm >= 0 && cv.pgemm <= 2);
  ISIC_CHECK_ARG(N > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Kh > 0 && Kw > 0 && up > 0);
  ISIC_CHECK_ARG(down == 1 || down == 2);
  ISIC_CHECK_ARG((stat_sum == nullptr) == (stat_sumsq == nullptr));
  ISIC_CHECK_ARG(!stat_sum || (stat_slots > 0 && down == 1));
  if (Cin % 64 != 0 || Cout % 64 != 0) return ISIC_ERR_UNSUPPORTED;
  // element offsets inside the kernels are 32-bit: both whole tensors must stay below 2^31 elements
  if ((int64_t)N * Hout * Wout * Cout > 0x7FFFFFFFLL || (int64_t)N * Hin * Win * Cin > 0x7FFFFFFFLL) return ISIC_ERR_UNSUPPORTED;
  const bool same3x3 = Kh == 3 && Kw == 3 && up == 1 && down == 1 && pad == 1 && Hin == Hout && Win == Wout;
  if (Cin == 64 && Cout == 64 && same3x3 && cv.c64 != 0) {
    const int rc = isic_conv3x3_c64_launch(cv.c64, in, w, out, N, Hin, Win, addend, addend_mask, stat_sum, stat_sumsq,
                                           stat_slots, cv.exp, as_stream(stream));
    return rc != ISIC_OK ? rc : isic_launch_status();
  }
  // >= 128-channel 3x3 layers: every input pixel staged once for all nine taps (conv_halo.hip)
  if (same3x3 && cv.halo != 1 && !(stat_sum && addend) && isic_conv_halo_supported(N, Hin, Win, Cin, Cout) &&
      (cv.halo == 2 || Cin >= 128)) {
    const int rc = isic_conv_halo_launch(in, w, out, N, Hin, Win, Cin, Cout, addend, addend_mask, stat_sum, stat_sumsq,
                                         stat_slots, nullptr, nullptr, cv.exp, as_stream(stream));
    return rc != ISIC_OK ? rc : isic_launch_status();
  }
  if (addend_mask) return ISIC_ERR_UNSUPPORTED;            // only the two pixels-staged-once kernels take a masked addend
  ConvArgs a;
  a.in = in; a.w = w; a.out = out; a.addend = addend;
  a.stat_sum = stat_sum; a.stat_sumsq = stat_sumsq; a.stat_slots = stat_slots > 0 ? stat_slots : 1;
  a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
  a.Kh = Kh; a.Kw = Kw; a.up = up; a.down_shift = down == 1 ? 0 : 1; a.pad = pad;
  a.ctiles = Cin / 64;
  hipStream_t s = as_stream(stream);
  // parity classes of the output grid (1 class for down == 1): all of them in one launch
  ConvArgsN all;
  all.n = 0;
  for (int ph = 0; ph < down; ++ph)
    for (int pw = 0; pw < down; ++pw) {
      a.ostep = down; a.oh0 = ph; a.ow0 = pw;
      a.Hs = (Hout - ph + down - 1) / down; a.Ws = (Wout - pw + down - 1) / down;
      if (a.Hs <= 0 || a.Ws <= 0) continue;
      a.kstep = down;
      // first tap with (o*up + k - pad) divisible by `down` (up == 1 when down > 1)
      a.kh0 = down == 1 ? 0 : (((pad - ph * up) % down) + down) % down;
      a.kw0 = down == 1 ? 0 : (((pad - pw * up) % down) + down) % down;
      a.nkh = a.kh0 < Kh ? (Kh - a.kh0 + down - 1) / down : 0;
      a.nkw = a.kw0 < Kw ? (Kw - a.kw0 + down - 1) / down : 0;
      a.M = N * a.Hs * a.Ws;
      if (a.M >= (1 << 24) || a.Hs * a.Ws >= (1 << 16)) return ISIC_ERR_UNSUPPORTED;   // fastdiv40 domain
      a.magic_hw = ((1ULL << 40) / (unsigned long long)(a.Hs * a.Ws)) + 1;
      a.magic_w = ((1ULL << 40) / (unsigned long long)a.Ws) + 1;
      a.Kmain = a.nkh * a.nkw * a.ctiles;
      a.Ktiles = a.Kmain;
      a.in2 = nullptr; a.w2 = nullptr;
      if (in2 && a.nkh == 1 && a.nkw == 1 && ph == 0 && pw == 0) {       // the class of the even output pixels: 1x1 / stride-2 taps land here
        a.in2 = in2; a.w2 = w2;
        a.Ktiles = a.Kmain + a.ctiles;
      }
      // a class without any tap still has to write (addend or zeros): Ktiles == 0 is handled by the kernel
      if (a.nkh > 16 || a.nkw > 16) return ISIC_ERR_UNSUPPORTED;   // per-row tap validity lives in 2 x 16+ bits
      all.c[all.n++] = a;
    }
  if (all.n == 0) return ISIC_OK;
  // short-K layers (stride-2 3x3 and its data gradient, 1x1 downsample): persistent blocks, register epilogue
  // (64-wide slices exist in conv_pgemm.hip but are not chosen: the data gradient of the 64 -> 128 stride-2 layer has 2-8
  //  K-tiles per tile and ran 0.73 ms there against 0.63 ms in the one-tile-per-block kernel, whose blocks overlap)
  //  Round 4: the PAIR gradient of that layer (a second source: 2-8 -> 4-8 K-tiles per tile, no addend to read) does run
  //  faster there: 0.66 against 0.82 ms at 2048 images (tools/halo_ab.py --pair), bit-equal.
  if (Cout % 64 == 0 && !(stat_sum && addend) && cv.pgemm != 1 &&
      (cv.pgemm == 2 || in2 != nullptr ||
       (Cout % 128 == 0 && (up != 1 || (down != 1 && Kh * Kw > 1) || (Kh == 1 && Kw == 1 && down == 1))))) {
    const int rc = isic_conv_pgemm_launch(all, s);
    return rc != ISIC_OK ? rc : isic_launch_status();
  }
  {
    int rc;
    const int mode = cv.mode;
    if (in2 && mode < 4) return ISIC_ERR_UNSUPPORTED;      // the second source is read by the LDS-DMA staging waves only
    if (Cout % 128 != 0) rc = launch_conv_mode<256, 64, 4, 1>(mode == 5 ? 4 : mode, all, s);
    else if (mode == 5) rc = launch_conv<256, 128, 4, 2, 5>(all, s);        // 8 MFMA + 8 staging waves, three stages
    else rc = launch_conv_mode<128, 128, 2, 2>(mode, all, s);
    if (rc != ISIC_OK) return rc;
  }
  return isic_launch_status();
}

}  // namespace

extern "C" {

int isic_conv2d_igemm_bf16(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int Hin, int Win, int Cin,
                           int Hout, int Wout, int Cout, int Kh, int Kw, int up, int down, int pad,
                           const uint16_t* addend, double* stat_sum, double* stat_sumsq, int stat_slots, void* stream) {
  return conv2d_dispatch(in, w, out, N, Hin, Win, Cin, Hout, Wout, Cout, Kh, Kw, up, down, pad, addend, stat_sum,
                         stat_sumsq, stat_slots, 0, stream);
}

int isic_test_conv2d_igemm_variant_bf16(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int Hin, int Win,
                                        int Cin, int Hout, int Wout, int Cout, int Kh, int Kw, int up, int down, int pad,
                                        const uint16_t* addend, double* stat_sum, double* stat_sumsq, int stat_slots,
                                        int variant, void* stream) {
  return conv2d_dispatch(in, w, out, N, Hin, Win, Cin, Hout, Wout, Cout, Kh, Kw, up, down, pad, addend, stat_sum,
                         stat_sumsq, stat_slots, variant, stream);
}

int isic_conv2d_dgrad_pair_bf16(const uint16_t* dy, const uint16_t* w, const uint16_t* dy2, const uint16_t* w2, uint16_t* dx,
                                int N, int Ho, int Wo, int Co, int H, int W, int C, void* stream) {
  ISIC_CHECK_ARG(dy && w && dy2 && w2 && dx);
  return conv2d_dispatch(dy, w, dx, N, Ho, Wo, Co, H, W, C, 3, 3, 1, 2, 1, nullptr, nullptr, nullptr, 0, 0, stream, nullptr,
                         dy2, w2);
}

int isic_test_conv2d_dgrad_pair_variant_bf16(const uint16_t* dy, const uint16_t* w, const uint16_t* dy2, const uint16_t* w2,
                                             uint16_t* dx, int N, int Ho, int Wo, int Co, int H, int W, int C, int variant,
                                             void* stream) {
  ISIC_CHECK_ARG(dy && w && dy2 && w2 && dx);
  return conv2d_dispatch(dy, w, dx, N, Ho, Wo, Co, H, W, C, 3, 3, 1, 2, 1, nullptr, nullptr, nullptr, 0, variant, stream,
                         nullptr, dy2, w2);
}

size_t isic_conv2d_maskadd_supported(int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int Kh, int Kw, int up,
                                    int down, int pad) {
  const bool same3x3 = Kh == 3 && Kw == 3 && up == 1 && down == 1 && pad == 1 && Hin == Hout && Win == Wout;
  if (!same3x3 || N <= 0) return 0;
  if (Cin == 64 && Cout == 64) return 1;
  return Cin >= 128 && isic_conv_halo_supported(N, Hin, Win, Cin, Cout) ? 1 : 0;
}

int isic_conv2d_igemm_maskadd_bf16(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int Hin, int Win, int Cin,
                                   int Hout, int Wout, int Cout, int Kh, int Kw, int up, int down, int pad,
                                   const uint16_t* addend, const uint8_t* addend_mask, void* stream) {
  ISIC_CHECK_ARG(addend && addend_mask);
  if (!isic_conv2d_maskadd_supported(N, Hin, Win, Cin, Hout, Wout, Cout, Kh, Kw, up, down, pad)) return ISIC_ERR_UNSUPPORTED;
  return conv2d_dispatch(in, w, out, N, Hin, Win, Cin, Hout, Wout, Cout, Kh, Kw, up, down, pad, addend, nullptr, nullptr, 0, 0,
                         stream, addend_mask);
}

int isic_conv_weight_prep_bf16(const float* w_krsc, uint16_t* w_fwd, uint16_t* w_dgrad, int O, int I, int Kh, int Kw,
                               void* stream) {
  ISIC_CHECK_ARG(w_krsc && (w_fwd || w_dgrad) && O > 0 && I > 0 && Kh > 0 && Kw > 0);
  const int64_t n = (int64_t)O * I * Kh * Kw;
  int64_t grid = (n + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(weight_prep_kernel, dim3((int)grid), dim3(256), 0, as_stream(stream), w_krsc, w_fwd, w_dgrad, O, I,
                     Kh, Kw);
  return isic_launch_status();
}

// Data gradient of a 3x3 / stride-1 convolution whose OUTPUT is the gradient of a ReLU(BatchNorm(y)) activation (+ an
// optional residual-gradient addend): see conv_halo.hip, STATS 2.  Geometry arguments as isic_conv2d_igemm_bf16.
size_t isic_conv2d_dgrad_bnbwd_supported(int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int Kh, int Kw,
                                      int up, int down, int pad) {
  const bool same3x3 = Kh == 3 && Kw == 3 && up == 1 && down == 1 && pad == 1 && Hin == Hout && Win == Wout;
  // (the 64 -> 64 kernel of conv_c64.hip keeps its weights in 144 of its 256 VGPRs: the same epilogue there spilled 10-28
  //  registers and cost 1.7 ms per step against 1.5 ms of reduction passes saved -- measured, not shipped)
  if (!same3x3 || N <= 0) return 0;
  return Cin >= 128 && isic_conv_halo_supported(N, Hin, Win, Cin, Cout) ? 1 : 0;
}

int isic_conv2d_dgrad_bnbwd_bf16(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int Hin, int Win, int Cin,
                                 int Hout, int Wout, int Cout, int Kh, int Kw, int up, int down, int pad,
                                 const uint16_t* addend, const uint8_t* relu_mask, const uint16_t* y_raw, double* sum_dz,
                                 double* sum_dzy, int stat_slots, void* stream) {
  ISIC_CHECK_ARG(in && w && out && relu_mask && y_raw && sum_dz && sum_dzy && stat_slots > 0);
  if (!isic_conv2d_dgrad_bnbwd_supported(N, Hin, Win, Cin, Hout, Wout, Cout, Kh, Kw, up, down, pad)) return ISIC_ERR_UNSUPPORTED;
  const int rc = isic_conv_halo_launch(in, w, out, N, Hin, Win, Cin, Cout, addend, nullptr, sum_dz, sum_dzy, stat_slots,
                                       relu_mask, y_raw, 0, as_stream(stream));
  return rc != ISIC_OK ? rc : isic_launch_status();
}

}  // extern "C"
