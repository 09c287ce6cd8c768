// Persistent fp16 GEMM on MFMA (gfx950) for the ViT-S/16 patch encoder of BASELINE.json configs[4]:
//   C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ residual),   fp16 in / out, fp32 accumulate
// i.e. nn.Linear with the bias, the GELU of the MLP and the residual add of a transformer block (or the broadcast
// position embedding of the patch projection) fused into the epilogue.  M = images x 196 tokens (0.4 M rows at 2048
// images), K, N in {384, 768, 1152, 1536}: K % 64 == 0, N % 128 == 0.
//
// Same machine as conv_pgemm.hip without the convolution addressing: 1024 threads, waves 0-7 multiply (4 x 2 waves of
// 64 x 64 = 4 x 4 v_mfma_f32_16x16x32_f16 tiles, operand roles swapped so that a lane ends with consecutive output
// columns of one row), waves 8-15 only issue LDS-DMA (four 1 KB pieces of A rows and two of W rows per K-tile of 64,
// 16-byte chunks XOR-swizzled on the source side), three 48 KB stages, one s_barrier per K-tile, counted vmcnt.  A block
// is persistent over a contiguous range of 256-row tiles of one 128-column slice (W stays hot in L2, the staging
// waves run two K-tiles ahead across tile boundaries) and the epilogue is register-only: the W rows are permuted inside
// a stage so that a lane owns eight consecutive columns -> one 16-byte residual load and one 16-byte store per 8 values.
//
// The reference's encoder is an un-vendored ConvMAE conv-ViT run frozen under no_grad (save_latent.py:42-60); this is
// this build's definition of the ViT-S/16 named by BASELINE.json (oracle/vit.py restates it on the CPU in fp32).

#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int GM = 256, GN = 128;
constexpr int G_A = GM * 128, G_B = GN * 128, G_STAGE = G_A + G_B;   // bytes per K-tile
constexpr int G_NST = 3;
constexpr int G_PER_IT = 6;                 // DMAs per staging wave and K-tile
constexpr int G_MAXSPB = 4;                 // column slices a block may own
constexpr int G_LDS = G_NST * G_STAGE + 1024 + 2 * G_MAXSPB * 512 + GM * 8;   // ring | DMA scratch | biases | LN column sums | LN row stats

struct GemmF16Args {
  const unsigned short* A;      // [M][K]
  const unsigned short* W;      // [N][K]
  const float* bias;            // [N] or null
  const unsigned short* res;    // [M][N] (res_rows == 0) or [res_rows][N] broadcast over m % res_rows, or null
  unsigned short* C;            // [M][N]
  int M, N, K, Ktiles, act, res_rows, mtiles, tiles_per_block;
  int spb;                      // consecutive 128-column slices per block (blockIdx.y owns slices y*spb .. +spb)
  // LayerNorm folded into the product (LNF): A is the RAW input x[M][K] of a LayerNorm over K, W holds W.diag(gamma),
  // bias holds bias + W.beta, ln_c[n] = sum_k W'[n][k], and
  //   C[m][n] = act(rstd_m (acc[m][n] - mean_m ln_c[n]) + bias[n])
  // ln_stats: [M][2] = (mean, rstd) when ln_parts == 0, else [M][ln_parts][2] partial (sum, sum of squares) of the row,
  // added in index order (what the STATS epilogue of the producing product writes)
  const float* ln_c;
  const float* ln_stats;
  int ln_parts;
  float ln_eps;
  // STATS epilogue (SOUT): per row and (slice, wn) the sum and the sum of squares of the 64 rounded outputs this wave
  // holds -> stats_out[m][2 (N / 128)][2]: the LayerNorm statistics of C without another pass over it
  float* stats_out;
};

__device__ __attribute__((aligned(256))) unsigned char g_g16_zero_page[256];

__device__ __forceinline__ void g16_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned g16_pack2(float lo, float hi) {
  const f16x2 h = {(_Float16)lo, (_Float16)hi};                     // round to nearest even
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float g16_lo(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[0]; }
__device__ __forceinline__ float g16_hi(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[1]; }
// erf-GELU (nn.GELU default, timm).  libm's erff: a hand-rolled Abramowitz-Stegun 7.1.26 with an exact reciprocal was
// measured SLOWER (1.01 vs 0.93 ms for fc1; no GELU at all: 0.87 ms -- the layer is bound by its 1.2 GB output).
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }

template <int GELU, bool RES, bool LNF, bool SOUT>
__global__ __launch_bounds__(1024) void gemm_f16_kernel(GemmF16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  constexpr int off_scr = G_NST * G_STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // work items of a block: (row tile, slice) with the slice fastest -- the A tile just staged for slice j is re-read
  // from L2 for slice j + 1, and the blocks (x, 0..gridDim.y-1) share an XCD (host: gridDim.x % 8 == 0)
  const int spb = a.spb;
  const int nbase = blockIdx.y * spb * GN;
  const int t_begin = blockIdx.x * a.tiles_per_block;
  const int ntl = min(a.mtiles - t_begin, a.tiles_per_block) * spb;        // work items
  if (ntl <= 0) return;                                    // whole block: no barrier has been reached yet
  const int KT = a.Ktiles;
  const int total_it = ntl * KT;

  if (wave >= 8) {
    // =================================================================== staging waves
    const int sw = wave - 8;
    const int r8 = lane >> 3;
    const int gch = (lane & 7) ^ r8;                       // global 16-byte chunk this lane fetches (swizzle on the source)
    const unsigned char* zp = g_g16_zero_page + (lane & 7) * 16;
    const unsigned scr = lds0 + off_scr;
    // W: stage rows sr = 16*sw + 8*t + r8 (t = 0, 1) hold output column
    //   wn*64 + 32*(j>>1) + 8*(rho>>2) + 4*(j&1) + (rho&3)   with wn = sr>>6, j = (sr>>4)&3, rho = sr&15
    // so that a lane's results of MFMA tiles 2t', 2t'+1 are EIGHT CONSECUTIVE columns (conv_halo.hip)
    const int chan0 = (sw >> 2) * 64 + ((sw & 3) >> 1) * 32 + (r8 >> 2) * 8 + (sw & 1) * 4 + (r8 & 3);   // t = 0; t = 1: + 16
    const unsigned short* wrow = nullptr;

    const unsigned short* a_ptr[4];
    bool a_ok[4];
    auto tile_rows = [&](int item) {
      const int tl = item / spb, j = item - tl * spb;
      wrow = a.W + (size_t)(nbase + j * GN + chan0) * a.K + gch * 8;
      const int m0 = (t_begin + tl) * GM;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + 8 * (sw + 8 * i) + r8;
        a_ok[i] = m < a.M;
        a_ptr[i] = a.A + (size_t)(a_ok[i] ? m : 0) * a.K + gch * 8;
      }
    };
    auto issue = [&](int kt, int stage, bool live) {
      const unsigned sbase = lds0 + stage * G_STAGE;
      const int koff = kt * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const void* src = (live && a_ok[i]) ? (const void*)(a_ptr[i] + koff) : (const void*)zp;
        g16_glds16(src, live ? sbase + (unsigned)((sw + 8 * i) * 1024) : scr);
      }
      g16_glds16(live ? (const void*)(wrow + koff) : (const void*)zp, live ? sbase + G_A + sw * 2048 : scr);
      g16_glds16(live ? (const void*)(wrow + koff + (size_t)16 * a.K) : (const void*)zp, live ? sbase + G_A + sw * 2048 + 1024 : scr);
    };
    int itile = 0, ikt = 0;
    tile_rows(0);
    auto advance = [&]() {
      if (++ikt == KT) { ikt = 0; ++itile; if (itile < ntl) tile_rows(itile); }
    };
    issue(0, 0, true); advance();
    issue(ikt, 1, total_it > 1); advance();
    for (int it = 0; it < total_it; ++it) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G_PER_IT) : "memory");     // K-tile `it` has landed
      __builtin_amdgcn_s_barrier();
      issue(ikt, (it + 2) % G_NST, it + 2 < total_it);
      advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA may outlive the block's LDS allocation
  } else {
    // ===================================================================== MFMA waves
    const int fr = lane & 15, fg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned aoff0 = (unsigned)((wm * 64 + fr) * 128 + ((fg ^ (fr & 7)) << 4));
    const unsigned boff0 = (unsigned)(G_A + (wn * 64 + fr) * 128 + ((fg ^ (fr & 7)) << 4));
    // the block's biases in LDS (read back per 4 columns in the epilogue: no registers held across the K loop)
    float* bias_all = reinterpret_cast<float*>(smem + off_scr + 1024);
    float* cn_all = bias_all + G_MAXSPB * GN;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2* rowst = reinterpret_cast<f32x2*>(cn_all + G_MAXSPB * GN);     // [GM] (rstd, mean * rstd) of the current row tile
    if (tid < spb * GN) {
      bias_all[tid] = a.bias ? a.bias[nbase + tid] : 0.f;
      if (LNF) cn_all[tid] = a.ln_c[nbase + tid];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // visible after the first K-tile barrier

    int it = 0;
    for (int item = 0; item < ntl; ++item) {
      const int tl = item / spb, j = item - tl * spb;
      const int m0 = (t_begin + tl) * GM;
      const unsigned chan = (unsigned)(nbase + j * GN + wn * 64 + fg * 8);
      const float* bias_lds = bias_all + j * GN;
      const float* cn_lds = cn_all + j * GN;
      f32x4 acc[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
      for (int kt = 0; kt < KT; ++kt, ++it) {
        __builtin_amdgcn_s_barrier();
        if (LNF && kt == 0 && j == 0 && tid < GM) {
          // every wave is past the previous item's epilogue (it came through this barrier), none reads the new values before
          // the next one (host: Ktiles >= 2): waves 0-3 finalise the LayerNorm statistics of the tile's 256 rows
          const int m = min(m0 + tid, a.M - 1);
          float mean, rstd;
          if (a.ln_parts == 0) {
            const f32x2 mr = *reinterpret_cast<const f32x2*>(a.ln_stats + (size_t)m * 2);
            mean = mr[0]; rstd = mr[1];
          } else {
            const f32x2* pp = reinterpret_cast<const f32x2*>(a.ln_stats + (size_t)m * a.ln_parts * 2);
            float s1 = 0.f, s2 = 0.f;
            for (int q = 0; q < a.ln_parts; ++q) { const f32x2 v = pp[q]; s1 += v[0]; s2 += v[1]; }
            const float inv = 1.f / (float)a.K;
            mean = s1 * inv;
            rstd = rsqrtf(fmaxf(s2 * inv - mean * mean, 0.f) + a.ln_eps);
          }
          rowst[tid] = (f32x2){rstd, mean * rstd};
        }
        const unsigned char* st = smem + (it % G_NST) * G_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f16x8 af[4], bfr[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            af[i] = *reinterpret_cast<const f16x8*>(st + ((aoff0 ^ (unsigned)(ks << 6)) + i * 2048));
            bfr[i] = *reinterpret_cast<const f16x8*>(st + ((boff0 ^ (unsigned)(ks << 6)) + i * 2048));
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
      }

      // ---- register-only epilogue: lane (fg, fr) holds, for MFMA tiles (i, 2t) and (i, 2t+1), the eight consecutive
      //      columns n0 + wn*64 + 32t + 8fg + {0..7} of row m0 + wm*64 + i*16 + fr
      size_t off[4];
      bool valid[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        valid[i] = m < a.M;
        off[i] = (size_t)(valid[i] ? m : 0) * a.N + chan;
      }
      auto res_row = [&](int i, u32x4 (&r)[2]) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        const size_t roff = a.res_rows > 0 ? (size_t)((valid[i] ? m : 0) % a.res_rows) * a.N + chan : off[i];
#pragma unroll
        for (int t = 0; t < 2; ++t) r[t] = *reinterpret_cast<const u32x4*>(a.res + roff + t * 32);
      };
      f32x2 rst[4];
      if (LNF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rst[i] = rowst[wm * 64 + i * 16 + fr];
      }
      // A row is emitted whole (both 64-byte halves t), then adjacent lanes swap one half (isic_pair_rows) so that every store
      // instruction writes 8 rows x 128 B instead of 16 rows x 64 B: the stores of this epilogue stream at 5.4 instead of
      // 3.2 TB/s (tests/probes/probe_rw.hip) -- they were 40-50 % of a launch.
      float ps1 = 0.f, ps2 = 0.f;                      // SOUT: sums of the row being emitted
      auto make = [&](int i, int t, const u32x4& rv) -> u32x4 {
        u32x4 v;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + wn * 64 + fg * 8 + t * 32 + 4 * h);
          f32x4 c;
          if (LNF) {
            const f32x4 cn = *reinterpret_cast<const f32x4*>(cn_lds + wn * 64 + fg * 8 + t * 32 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = fmaf(acc[i][2 * t + h][e], rst[i][0], fmaf(-rst[i][1], cn[e], bv[e]));
          } else {
            c = acc[i][2 * t + h] + bv;
          }
          if (GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = gelu_erf(c[e]);
          }
          if (RES) {
            const unsigned lo = rv[2 * h], hi = rv[2 * h + 1];
            c[0] += g16_lo(lo); c[1] += g16_hi(lo); c[2] += g16_lo(hi); c[3] += g16_hi(hi);
          }
          const f16x2 p0 = {(_Float16)c[0], (_Float16)c[1]}, p1 = {(_Float16)c[2], (_Float16)c[3]};   // round to nearest even
          v[2 * h] = __builtin_bit_cast(unsigned, p0);
          v[2 * h + 1] = __builtin_bit_cast(unsigned, p1);
          if (SOUT) {                    // sums of the ROUNDED values: what a LayerNorm reading C would see
            const f16x2 one = {(_Float16)1.f, (_Float16)1.f};
            ps1 = __builtin_amdgcn_fdot2(p0, one, ps1, false);
            ps2 = __builtin_amdgcn_fdot2(p0, p0, ps2, false);
            ps1 = __builtin_amdgcn_fdot2(p1, one, ps1, false);
            ps2 = __builtin_amdgcn_fdot2(p1, p1, ps2, false);
          }
        }
        return v;
      };
      const bool odd = fr & 1;
      const int parts = 2 * (a.N / GN), part = 2 * ((nbase / GN) + j) + wn;
      u32x4 cur[2] = {}, nxt[2] = {};
      if (RES) res_row(0, cur);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (RES && i + 1 < 4) res_row(i + 1, nxt);                       // the next row's residual in flight
        ps1 = ps2 = 0.f;
        const u32x4 v0 = make(i, 0, cur[0]);
        if (GELU) __builtin_amdgcn_sched_barrier(0);                     // one half at a time: erff's temporaries x 16 values spill
        const u32x4 v1 = make(i, 1, cur[1]);
        if (RES) { cur[0] = nxt[0]; cur[1] = nxt[1]; }
        u32x4 da, db;
        isic_pair_rows(v0, v1, odd, da, db);
        const int mA = m0 + wm * 64 + i * 16 + (fr & ~1), mB = mA + 1;
        const size_t col = (size_t)chan + (odd ? 32 : 0);
        if (mA < a.M) __builtin_nontemporal_store(da, reinterpret_cast<u32x4*>(a.C + (size_t)mA * a.N + col));
        if (mB < a.M) __builtin_nontemporal_store(db, reinterpret_cast<u32x4*>(a.C + (size_t)mB * a.N + col));
        if (SOUT) {
          // the four fg lanes of a row hold its 64 columns of this wave: fixed-order butterfly, lane fg == 0 writes
          float s1 = ps1, s2 = ps2;
          s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
          const int m = m0 + wm * 64 + i * 16 + fr;
          if (fg == 0 && valid[i]) *reinterpret_cast<f32x2*>(a.stats_out + ((size_t)m * parts + part) * 2) = (f32x2){s1, s2};
        }
      }
    }
  }   // MFMA waves
}

template <int GELU, bool RES, bool LNF = false, bool SOUT = false>
int launch_g16(const GemmF16Args& a, dim3 grid, hipStream_t stream) {
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (isic_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<GELU, RES, LNF, SOUT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_f16_kernel<GELU, RES, LNF, SOUT>), grid, dim3(1024), G_LDS, stream, a);
  return isic_launch_status();
}

// grid + the fields of `a` that depend on it
dim3 g16_plan(GemmF16Args& a) {
  const int cus = isic_cu_count();
  const int M = a.M, N = a.N, K = a.K;
  a.Ktiles = K / 64;
  a.mtiles = (M + GM - 1) / GM;
  const int nslices = N / GN;
  // All N-slices of one row range must share an L2: workgroup w runs on XCD w % 8 and w = x + y * gx, so gx is a multiple
  // of 8 -- the blocks (x, 0..gy-1) then sit on one XCD and walk the same A tiles in step -- and a block owns `spb`
  // consecutive slices (slice fastest), chosen so that gx * gy fills the chip: 12 slices -> 3 per block x 4 x 64 blocks,
  // 9 -> 3 x 3 x 80, 3 -> 1 x 3 x 80 (qkv 652 -> 569 us, fc2 691 -> 629 us at 2048 images).  A is then read from HBM about
  // once instead of once per slice.
  int spb = 1, gx = 1, best = -1;
  for (int c = 1; c <= G_MAXSPB; ++c) {
    if (nslices % c != 0) continue;
    const int gy_c = nslices / c;
    int gx_c = cus / gy_c;
    if (gx_c >= 8) gx_c &= ~7;
    if (gx_c < 1) gx_c = 1;
    if (gx_c > a.mtiles) gx_c = a.mtiles;
    // a block re-reads its A tile one tile-time later: the A tiles an XCD touches meanwhile must fit its 4 MB L2
    if (c > 1 && (int64_t)((gx_c + 7) / 8) * GM * K * 2 > (3 << 20)) continue;
    if (gx_c * gy_c > best) { best = gx_c * gy_c; spb = c; gx = gx_c; }
  }
  const int gy = nslices / spb;
  a.spb = spb;
  a.tiles_per_block = (a.mtiles + gx - 1) / gx;
  return dim3(gx, gy);
}

}  // namespace

extern "C" {

int isic_gemm_f16(const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual, uint16_t* C,
                  int M, int N, int K, int act, int residual_rows, void* stream) {
  return isic_gemm_f16_stats(A, W, bias, residual, C, nullptr, M, N, K, act, residual_rows, stream);
}

int isic_gemm_f16_stats(const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* residual, uint16_t* C,
                        float* row_stats, int M, int N, int K, int act, int residual_rows, void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (act == 0 || act == 1) && residual_rows >= 0);
  if (M == 0) return ISIC_OK;
  ISIC_CHECK_ARG(A && W && C);
  ISIC_CHECK_ARG(residual || residual_rows == 0);
  if (N % GN != 0 || K % 64 != 0) return ISIC_ERR_UNSUPPORTED;
  GemmF16Args a;
  a.A = A; a.W = W; a.bias = bias; a.res = residual; a.C = C;
  a.M = M; a.N = N; a.K = K; a.act = act; a.res_rows = residual_rows;
  a.ln_c = nullptr; a.ln_stats = nullptr; a.ln_parts = 0; a.ln_eps = 0.f; a.stats_out = row_stats;
  const dim3 grid = g16_plan(a);
  hipStream_t s = as_stream(stream);
  if (row_stats) {                                     // the two producers of a LayerNorm input: residual products
    if (act == 1 || !residual) return ISIC_ERR_UNSUPPORTED;
    return launch_g16<0, true, false, true>(a, grid, s);
  }
  if (act == 1) return residual ? ISIC_ERR_UNSUPPORTED : launch_g16<1, false>(a, grid, s);   // GELU + residual: not a ViT layer
  return residual ? launch_g16<0, true>(a, grid, s) : launch_g16<0, false>(a, grid, s);
}

int isic_gemm_f16_ln(const uint16_t* X, const uint16_t* Wg, const float* bias_b, const float* ln_c, const float* ln_stats,
                     int ln_parts, uint16_t* C, int M, int N, int K, int act, float eps, void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (act == 0 || act == 1) && ln_parts >= 0 && eps >= 0.f);
  if (M == 0) return ISIC_OK;
  ISIC_CHECK_ARG(X && Wg && bias_b && ln_c && ln_stats && C);
  if (N % GN != 0 || K % 64 != 0 || K < 128) return ISIC_ERR_UNSUPPORTED;      // the row statistics need two K-tiles
  GemmF16Args a;
  a.A = X; a.W = Wg; a.bias = bias_b; a.res = nullptr; a.C = C;
  a.M = M; a.N = N; a.K = K; a.act = act; a.res_rows = 0;
  a.ln_c = ln_c; a.ln_stats = ln_stats; a.ln_parts = ln_parts; a.ln_eps = eps; a.stats_out = nullptr;
  const dim3 grid = g16_plan(a);
  hipStream_t s = as_stream(stream);
  return act == 1 ? launch_g16<1, false, true, false>(a, grid, s) : launch_g16<0, false, true, false>(a, grid, s);
}

}  // extern "C"
