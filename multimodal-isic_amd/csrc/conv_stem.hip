// ResNet stem convolution 7x7 / stride 2 / pad 3, 3 -> 64 channels, on bf16 MFMA (gfx950).
//
// Cin = 3 is too thin for the generic implicit GEMM (K-tiles of 64 channels), so the stem gets its
// own direct kernels.  The input is NHWC with C padded to 4 (8 bytes per pixel); a block stages the
// (21 x 40)-pixel input patch of an 8 x 16 output tile in LDS once and every MFMA A-fragment is a
// single ds_read_b128 out of it: for output pixel (oy,ox) and kernel row kh the 8 taps kw = 0..7
// (the 8th is a zero-weight pad) x 4 channels are 32 CONTIGUOUS bf16 of patch row 2*oy+kh starting
// at column 2*ox, so K = 7 steps of 32.  Weights are packed [64][7][8][4] (kw, c zero padded).
//
// The weight gradient uses the same patch: K = pixels, the dY tile is staged [pixel][co] and both
// operands are fetched with the transposing LDS read ds_read_b64_tr_b16 (for the input operand each
// lane addresses its own pixel's 4 taps x 4 channels = 8 contiguous bytes).
//
// No counterpart in the reference (encoder un-vendored, save_latent.py:42-60); ResNet-18 layer
// table: SURVEY.md 8d (conv1: 118 MMAC per 224x224 image).
#include <type_traits>

#include "common.h"
#include "pool_grad.h"

namespace {

constexpr int TH = 8, TW = 16;             // output tile
constexpr int PR = 2 * TH + 5, PC = 40;    // patch rows (21) / cols (38 used, 40 allocated)
constexpr int PROW = PC * 8;               // bytes per patch row (4 bf16 per pixel)
constexpr int WROW = 7 * 64 + 16;          // bytes per co row of the LDS weight image (padded: conflict-free)
constexpr int CPAD = 72;                   // epilogue row stride (elements)
constexpr int YROW = 128 + 32;             // bytes per pixel row of the staged dY tile
constexpr int STEM_DW_ELEMS = 64 * 7 * 7 * 3;   // the weight gradient, [co][kh][kw][c]
constexpr int STEM_WGRAD_BLOCKS = 512;
constexpr int STEM_FWD_BLOCKS = 768;       // persistent blocks of the forward kernel: three per CU (LDS, registers)     // persistent blocks of the weight-gradient kernels (one partial each)

struct StemArgs {
  const unsigned short* in;   // [N,Hin,Win,4]
  const unsigned short* w;    // [64][7][8][4]
  unsigned short* out;        // [N,Hout,Wout,64]
  const unsigned short* dy;   // wgrad: [N,Hout,Wout,64]
  float* dw;                  // wgrad: [64][7][7][3] fp32
  double* stat_sum;           // fwd, optional: [slots][64] BatchNorm sum / sum of squares of the rounded outputs
  double* stat_sumsq;
  int stat_slots;
  int N, Hin, Win, Hout, Wout, tiles_h, tiles_w, total_tiles;
};

// The input patch of a tile, staged in two halves -- global loads into registers, registers into LDS -- so that the
// loads of the NEXT tile are in flight while the current tile is multiplied (PR * PC = 840 pixels: at most 4 per thread).
// patch pixel (pr, pc) = input pixel (2*oy0 - 3 + pr, 2*ox0 - 3 + pc); 8 bytes each
constexpr int PATCH_PER_THREAD = (PR * PC + 255) / 256;
__device__ __forceinline__ void fetch_patch(u32x2 (&r)[PATCH_PER_THREAD], const StemArgs& a, int tile, int tid) {
  const int n = tile / (a.tiles_h * a.tiles_w);
  const int t2 = tile - n * (a.tiles_h * a.tiles_w);
  const int oy0 = (t2 / a.tiles_w) * TH, ox0 = (t2 % a.tiles_w) * TW;
#pragma unroll
  for (int u = 0; u < PATCH_PER_THREAD; ++u) {
    const int idx = tid + 256 * u;
    const int pr = idx / PC, pc = idx - pr * PC;
    const int hi = 2 * oy0 - 3 + pr, wi = 2 * ox0 - 3 + pc;
    u32x2 v = {0u, 0u};
    if (tile < a.total_tiles && idx < PR * PC && hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win)
      v = *reinterpret_cast<const u32x2*>(a.in + (((size_t)n * a.Hin + hi) * a.Win + wi) * 4);
    r[u] = v;
  }
}
__device__ __forceinline__ void commit_patch(unsigned char* Ps, const u32x2 (&r)[PATCH_PER_THREAD], int tid) {
#pragma unroll
  for (int u = 0; u < PATCH_PER_THREAD; ++u) {
    const int idx = tid + 256 * u;
    if (idx < PR * PC) {
      const int pr = idx / PC, pc = idx - pr * PC;
      *reinterpret_cast<u32x2*>(Ps + pr * PROW + pc * 8) = r[u];
    }
  }
}

__global__ __launch_bounds__(256) void conv_stem_fwd_kernel(StemArgs a) {
  // the C tile takes the patch's place once the tile is multiplied (one more barrier per tile): 48 KB of LDS and 156
  // registers = THREE blocks per CU instead of two (2.96 -> 2.76 ms at 4096 images: a tile is 6.7 KB in and 16 KB out
  // between three barriers, the kernel waits on those round trips, not on the matrix cores)
  static_assert(TH * TW * CPAD * 2 >= PR * PROW, "the C tile overlays the patch");
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * WROW + TH * TW * CPAD * 2];
  unsigned char* Ws = smem;
  unsigned char* Ps = smem + 64 * WROW;
  unsigned short* Cs = reinterpret_cast<unsigned short*>(smem + 64 * WROW);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;

  // weights -> LDS once per block: 64 rows x 448 bytes (28 chunks of 16 B)
  for (int idx = tid; idx < 64 * 28; idx += 256) {
    const int co = idx / 28, ch = idx - co * 28;
    *reinterpret_cast<u32x4*>(Ws + co * WROW + ch * 16) = *reinterpret_cast<const u32x4*>(a.w + co * 224 + ch * 8);
  }

  u32x2 pre[PATCH_PER_THREAD];
  fetch_patch(pre, a, blockIdx.x, tid);
  // fused BatchNorm statistics of the ROUNDED outputs, in registers over all tiles of the block: lane (fg, fi) owns
  // channels 16j + 4fg + r of pixel column fi -- 16 sums + 16 sums of squares (fp32; ~800 values each), reduced over the
  // 16 lanes of a DPP row and the four waves once, at the end (the LDS read-back loop this replaces cost 0.3 ms per step)
  float st_s[4][4], st_q[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st_s[j][r] = 0.f; st_q[j][r] = 0.f; }
  for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
    const int n = tile / (a.tiles_h * a.tiles_w);
    const int t2 = tile - n * (a.tiles_h * a.tiles_w);
    const int oy0 = (t2 / a.tiles_w) * TH, ox0 = (t2 % a.tiles_w) * TW;
    lds_barrier();    // previous tile's patch and C tile fully consumed / weights visible (no wait for its stores)
    commit_patch(Ps, pre, tid);
    lds_barrier();                                       // (NOT __syncthreads(): that drains vmcnt, i.e. waits for the previous
                                                         //  tile's output stores to complete their round trip to memory)
    fetch_patch(pre, a, tile + gridDim.x, tid);          // next tile's loads fly under this tile's MFMAs and stores

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
      bf16x8 af[2], bfr[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int oyl = wave * 2 + i;
        af[i] = *reinterpret_cast<const bf16x8*>(Ps + (2 * oyl + kh) * PROW + (2 * fi + 2 * fg) * 8);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(Ws + (j * 16 + fi) * WROW + kh * 64 + fg * 16);
      // swapped operand roles, D[co][pixel]: lane (fg, fi) ends with output channels 16j + 4fg + {0..3} of pixel fi
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    lds_barrier();                                       // every wave is done with the patch: the C tile overwrites it
    // epilogue through LDS: pixel index p = oyl*16 + oxl, row-major [p][co]; one packed 8-byte store per MFMA tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = (wave * 2 + i) * 16 + fi;
        u32x2 v;
        v[0] = (unsigned)f32_to_bf16_bits(acc[i][j][0]) | ((unsigned)f32_to_bf16_bits(acc[i][j][1]) << 16);
        v[1] = (unsigned)f32_to_bf16_bits(acc[i][j][2]) | ((unsigned)f32_to_bf16_bits(acc[i][j][3]) << 16);
        *reinterpret_cast<u32x2*>(Cs + p * CPAD + j * 16 + fg * 4) = v;
        if (a.stat_sum) {
          const bool in = (oy0 + wave * 2 + i < a.Hout) && (ox0 + fi < a.Wout);
          const float r0 = in ? __uint_as_float(v[0] << 16) : 0.f, r1 = in ? __uint_as_float(v[0] & 0xFFFF0000u) : 0.f;
          const float r2 = in ? __uint_as_float(v[1] << 16) : 0.f, r3 = in ? __uint_as_float(v[1] & 0xFFFF0000u) : 0.f;
          st_s[j][0] += r0; st_q[j][0] += r0 * r0;
          st_s[j][1] += r1; st_q[j][1] += r1 * r1;
          st_s[j][2] += r2; st_q[j][2] += r2 * r2;
          st_s[j][3] += r3; st_q[j][3] += r3 * r3;
        }
      }
    lds_barrier();
    for (int idx = tid; idx < TH * TW * 8; idx += 256) {
      const int p = idx >> 3, ch = idx & 7;
      const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
      if (oy < a.Hout && ox < a.Wout)
        __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(Cs + p * CPAD + ch * 8),
                                    reinterpret_cast<u32x4*>(a.out + (((size_t)n * a.Hout + oy) * a.Wout + ox) * 64 + ch * 8));
    }
  }
  if (a.stat_sum) {
    // DPP row sums over the 16 pixel columns, then the four waves meet in LDS (the weight image is dead by now)
    __shared__ float stat_red[2][4][64];
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sv = st_s[j][r], qv = st_q[j][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { sv += __shfl_xor(sv, o, 16); qv += __shfl_xor(qv, o, 16); }
        if (fi == 0) { stat_red[0][wave][j * 16 + fg * 4 + r] = sv; stat_red[1][wave][j * 16 + fg * 4 + r] = qv; }
      }
    lds_barrier();
    if (tid < 128) {
      const int c = tid & 63, w = tid >> 6;
      const float t = (stat_red[w][0][c] + stat_red[w][1][c]) + (stat_red[w][2][c] + stat_red[w][3][c]);
      const size_t slot = (size_t)(blockIdx.x % a.stat_slots) * 64 + c;
      atomicAdd((w == 0 ? a.stat_sum : a.stat_sumsq) + slot, (double)t);
    }
  }
}

__global__ __launch_bounds__(256) void conv_stem_wgrad_kernel(StemArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[PR * PROW + TH * TW * YROW];
  unsigned char* Ps = smem;
  unsigned char* Ys = smem + PR * PROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fg = lane >> 4, fi = lane & 15, fq = fi >> 2, fp = fi & 3;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

  // n-tiles: nt = kh*2 + half (14 of them); wave w owns nt = w, w+4, w+8, w+12
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // operands of the next tile are fetched into registers while the current one is multiplied
  u32x2 pre[PATCH_PER_THREAD];
  u32x4 prey[4];
  auto fetch_dy = [&](int tile) {
    const int n = tile / (a.tiles_h * a.tiles_w);
    const int t2 = tile - n * (a.tiles_h * a.tiles_w);
    const int oy0 = (t2 / a.tiles_w) * TH, ox0 = (t2 % a.tiles_w) * TW;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;
      const int p = idx >> 3, ch = idx & 7;
      const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (tile < a.total_tiles && oy < a.Hout && ox < a.Wout)
        v = *reinterpret_cast<const u32x4*>(a.dy + (((size_t)n * a.Hout + oy) * a.Wout + ox) * 64 + ch * 8);
      prey[u] = v;
    }
  };
  fetch_patch(pre, a, blockIdx.x, tid);
  fetch_dy(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
    lds_barrier();                                       // previous tile's operands fully consumed
    commit_patch(Ps, pre, tid);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;
      *reinterpret_cast<u32x4*>(Ys + (idx >> 3) * YROW + (idx & 7) * 16) = prey[u];
    }
    lds_barrier();
    fetch_patch(pre, a, tile + gridDim.x, tid);
    fetch_dy(tile + gridDim.x);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int k1 = ks * 32 + 8 * fg + fq, k2 = k1 + 4;   // the two pixel rows this lane addresses
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Ys + k1 * YROW + (i * 16 + 4 * fp) * 2));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Ys + k2 * YROW + (i * 16 + 4 * fp) * 2));
        s16x8_t t; t.lo = lo; t.hi = hi;
        af[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nt = wave + 4 * j;
        const int kh = nt >> 1, half = nt & 1;
        // nt >= 14 would address past the taps: clamp the address (result discarded at the end)
        const int khc = kh > 6 ? 6 : kh;
        const unsigned char* p1 = Ps + (2 * (k1 >> 4) + khc) * PROW + (2 * (k1 & 15) + half * 4 + fp) * 8;
        const unsigned char* p2 = Ps + (2 * (k2 >> 4) + khc) * PROW + (2 * (k2 & 15) + half * 4 + fp) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p1);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p2);
        s16x8_t t; t.lo = lo; t.hi = hi;
        bfr[j] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  // dw[co][kh][kw][c], c < 3, kw < 7
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nt = wave + 4 * j;
      if (nt >= 14) continue;
      const int kh = nt >> 1, kw = (nt & 1) * 4 + (fi >> 2), c = fi & 3;
      if (kw >= 7 || c >= 3) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = i * 16 + fg * 4 + r;
        // the block's OWN partial (workspace [grid][64*7*7*3]); stem_wgrad_reduce_kernel adds them in block order:
        // deterministic, where fp32 atomics added the blocks in arrival order
        a.dw[(size_t)blockIdx.x * STEM_DW_ELEMS + ((co * 7 + kh) * 7 + kw) * 3 + c] = acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------- weight gradient fed by the POOLED gradient
// Same tiles and MFMA schedule as conv_stem_wgrad_kernel, but the dY tile is never read from memory: it is the
// BatchNorm(+ReLU) backward of the 3x3/2 max-pool backward of the pooled gradient, formed in registers from the stem
// activation y0 (read once, here), the pooled gradient and the argmax codes -- what stem_bn_bwd_apply_kernel would have
// written (3.3 GB at 2048 images) and this kernel read back.  A thread owns the 2x2 pixels (2a+dy, 2b+dx) of one
// 8-channel group (pool_grad.h): 8 x 16 tile = 4 x 8 such blocks x 8 groups = 256 threads.  The raw loads of the NEXT
// tile stay in registers while the current one is multiplied; the arithmetic happens when they are committed to LDS.
// VALU-bound: ~1,650 vector instructions per thread and tile (max-pool routing: 9 compare / select / add per channel;
// BatchNorm: ~8 per value; addressing) = 6.6 k cycles per tile and SIMD against 1 k of MFMA.  Measured and rejected:
// a producer / consumer split (4 waves load + form dY with two tiles of operands in flight, 4 waves multiply: 2.7 ms
// vs 2.15 ms -- one VALU wave per SIMD issues worse than two) and wave-uniform base + per-thread constant offsets with an
// incremental tile cursor (2.11 ms: the address arithmetic was not the bulk of it).
struct StemBnArgs {
  StemArgs s;                       // s.dy unused
  const unsigned short* y0;         // [N,Hout,Wout,64] stem convolution output
  const unsigned char* argmax;      // [N,Hp,Wp,64]
  const unsigned short* gp;         // [N,Hp,Wp,64] gradient of the pooled activation
  const float* mean; const float* rstd; const float* gamma; const float* scale; const float* shift;
  const double* dgamma; const double* dbeta;
  float* dgamma_f32; float* dbeta_f32;
  int Hp, Wp;
};

__global__ __launch_bounds__(256, 2) void conv_stem_wgrad_bn_kernel(StemBnArgs b) {
  const StemArgs& a = b.s;
  __shared__ __attribute__((aligned(16))) unsigned char smem[PR * PROW + TH * TW * YROW];
  __shared__ __attribute__((aligned(16))) float cst[5][64];     // A, B, D of dY = A*dz + B*y0 + D; sc, sh of the ReLU mask
  unsigned char* Ps = smem;
  unsigned char* Ys = smem + PR * PROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fg = lane >> 4, fi = lane & 15, fq = fi >> 2, fp = fi & 3;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const isic_pool::PoolGeom geom{a.Hout, a.Wout, 64, b.Hp, b.Wp};

  if (tid < 64) {
    const float inv_rows = 1.f / (float)((int64_t)a.N * a.Hout * a.Wout);
    // dY = k1*(dz - k2 - xh*k3), xh = (y0 - mu)*rs  ==  A*dz + B*y0 + D  (three constants, two FMAs per element)
    const float rs = b.rstd[tid], mu = b.mean[tid], k1 = b.gamma[tid] * rs;
    const float k2 = (float)b.dbeta[tid] * inv_rows, k3 = (float)b.dgamma[tid] * inv_rows;
    cst[0][tid] = k1; cst[1][tid] = -k1 * k3 * rs; cst[2][tid] = k1 * (k3 * rs * mu - k2);
    cst[3][tid] = b.scale[tid]; cst[4][tid] = b.shift[tid];
    if (blockIdx.x == 0 && b.dgamma_f32) {
      b.dgamma_f32[tid] += (float)b.dgamma[tid];
      b.dbeta_f32[tid] += (float)b.dbeta[tid];
    }
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this thread's 2x2 pixel block inside the tile and its channel group
  const int cg = tid & 7, bl = tid >> 3, al = bl >> 3, bw = bl & 7;
  u32x2 pre[PATCH_PER_THREAD];
  u32x4 xr[4];
  isic_pool::Windows win;
  auto fetch_dy = [&](int tile) {
    const int tc = tile < a.total_tiles ? tile : a.total_tiles - 1;       // clamped: always valid addresses
    const int n = tc / (a.tiles_h * a.tiles_w);
    const int t2 = tc - n * (a.tiles_h * a.tiles_w);
    const int oy0 = (t2 / a.tiles_w) * TH, ox0 = (t2 % a.tiles_w) * TW;
    const int ga = (oy0 >> 1) + al, gb = (ox0 >> 1) + bw;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int hi = min(2 * ga + (p >> 1), a.Hout - 1), wi = min(2 * gb + (p & 1), a.Wout - 1);
      xr[p] = *reinterpret_cast<const u32x4*>(b.y0 + (((size_t)n * a.Hout + hi) * a.Wout + wi) * 64 + cg * 8);
    }
    isic_pool::load_windows(win, b.argmax, b.gp, geom, n, min(ga, b.Hp - 1), min(gb, b.Wp - 1), cg);
  };
  // raw loads -> dY values of the 2x2 pixels -> LDS rows [pixel][co]
  auto commit_dy = [&](int tile) {
    const int n = tile / (a.tiles_h * a.tiles_w);
    const int t2 = tile - n * (a.tiles_h * a.tiles_w);
    const int oy0 = (t2 / a.tiles_w) * TH, ox0 = (t2 % a.tiles_w) * TW;
    const int ga = (oy0 >> 1) + al, gb = (ox0 >> 1) + bw;
    const bool inside = (oy0 + TH <= a.Hout) && (ox0 + TW <= a.Wout);    // block-uniform: the usual case
    auto half = [&](auto HC) {
      constexpr int H_ = decltype(HC)::value;
      float g[4][4];
      isic_pool::windows_to_grad4<H_>(win, geom, ga, gb, g);
      const int c0 = cg * 8 + H_ * 4;
      const f32x4 kA = *reinterpret_cast<const f32x4*>(&cst[0][c0]), kB = *reinterpret_cast<const f32x4*>(&cst[1][c0]);
      const f32x4 kD = *reinterpret_cast<const f32x4*>(&cst[2][c0]), sc = *reinterpret_cast<const f32x4*>(&cst[3][c0]);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(&cst[4][c0]);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int oyl = 2 * al + (p >> 1), oxl = 2 * bw + (p & 1);
        const bool in = inside || ((oy0 + oyl < a.Hout) && (ox0 + oxl < a.Wout));
        const unsigned lo = xr[p][2 * H_], hi = xr[p][2 * H_ + 1];
        const float xv[4] = {__uint_as_float(lo << 16), __uint_as_float(lo & 0xFFFF0000u), __uint_as_float(hi << 16),
                             __uint_as_float(hi & 0xFFFF0000u)};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float dz = (xv[j] * sc[j] + sh[j] > 0.f) ? g[p][j] : 0.f;
          o[j] = kA[j] * dz + (kB[j] * xv[j] + kD[j]);
          if (!inside) o[j] = in ? o[j] : 0.f;
        }
        u32x2 v;
        v[0] = (unsigned)f32_to_bf16_bits(o[0]) | ((unsigned)f32_to_bf16_bits(o[1]) << 16);
        v[1] = (unsigned)f32_to_bf16_bits(o[2]) | ((unsigned)f32_to_bf16_bits(o[3]) << 16);
        *reinterpret_cast<u32x2*>(Ys + (oyl * TW + oxl) * YROW + cg * 16 + H_ * 8) = v;
      }
    };
    half(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);                   // keep the two halves' temporaries from overlapping
    half(std::integral_constant<int, 1>{});
  };
  __syncthreads();                                       // constants visible
  fetch_patch(pre, a, blockIdx.x, tid);
  fetch_dy(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
    lds_barrier();                                       // previous tile's operands fully consumed
    commit_patch(Ps, pre, tid);
    commit_dy(tile);
    lds_barrier();
    fetch_patch(pre, a, tile + gridDim.x, tid);
    fetch_dy(tile + gridDim.x);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int k1 = ks * 32 + 8 * fg + fq, k2 = k1 + 4;   // the two pixel rows this lane addresses
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Ys + k1 * YROW + (i * 16 + 4 * fp) * 2));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Ys + k2 * YROW + (i * 16 + 4 * fp) * 2));
        s16x8_t t; t.lo = lo; t.hi = hi;
        af[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nt = wave + 4 * j;
        const int kh = nt >> 1, half = nt & 1;
        const int khc = kh > 6 ? 6 : kh;
        const unsigned char* p1 = Ps + (2 * (k1 >> 4) + khc) * PROW + (2 * (k1 & 15) + half * 4 + fp) * 8;
        const unsigned char* p2 = Ps + (2 * (k2 >> 4) + khc) * PROW + (2 * (k2 & 15) + half * 4 + fp) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p1);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p2);
        s16x8_t t; t.lo = lo; t.hi = hi;
        bfr[j] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nt = wave + 4 * j;
      if (nt >= 14) continue;
      const int kh = nt >> 1, kw = (nt & 1) * 4 + (fi >> 2), c = fi & 3;
      if (kw >= 7 || c >= 3) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = i * 16 + fg * 4 + r;
        // the block's OWN partial (workspace [grid][64*7*7*3]); stem_wgrad_reduce_kernel adds them in block order:
        // deterministic, where fp32 atomics added the blocks in arrival order
        a.dw[(size_t)blockIdx.x * STEM_DW_ELEMS + ((co * 7 + kh) * 7 + kw) * 3 + c] = acc[i][j][r];
      }
    }
}

// dw[e] += sum over the blocks' partials, in block order: 16 lanes per element (lane g takes blocks g, g + 16, ...) joined
// by a fixed xor tree
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                 int nparts) {
  const int g = threadIdx.x & 15;
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4);
  float v = 0.f;
  if (e < STEM_DW_ELEMS)
    for (int s = g; s < nparts; s += 16) v += partial[(size_t)s * STEM_DW_ELEMS + e];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
  if (e < STEM_DW_ELEMS && g == 0) dw[e] += v;
}

// fp32 [64][7][7][3] (channels_last memory of the OIHW parameter) -> bf16 [64][7][8][4], zero padded
__global__ void stem_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ ws) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 64 * 7 * 8 * 4) return;
  const int c = idx & 3, kw = (idx >> 2) & 7, kh = (idx >> 5) % 7, co = idx / 224;
  float v = 0.f;
  if (c < 3 && kw < 7) v = w[((co * 7 + kh) * 7 + kw) * 3 + c];
  ws[idx] = f32_to_bf16_bits(v);
}

int stem_args(StemArgs& a, int N, int Hin, int Win, int Hout, int Wout) {
  if (Hout != (Hin + 6 - 7) / 2 + 1 || Wout != (Win + 6 - 7) / 2 + 1) return ISIC_ERR_BAD_ARG;
  a.N = N; a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout;
  a.tiles_h = ceil_div(Hout, TH); a.tiles_w = ceil_div(Wout, TW);
  const int64_t tt = (int64_t)N * a.tiles_h * a.tiles_w;
  if (tt > 0x7FFFFFFFLL) return ISIC_ERR_UNSUPPORTED;
  a.total_tiles = (int)tt;
  return ISIC_OK;
}

}  // namespace

extern "C" {

int isic_conv_stem_fwd_bf16(const uint16_t* in_nhwc4, const uint16_t* w_stem, uint16_t* out, int N, int Hin, int Win,
                            int Hout, int Wout, void* stream) {
  return isic_conv_stem_fwd_stats_bf16(in_nhwc4, w_stem, out, N, Hin, Win, Hout, Wout, nullptr, nullptr, 0, stream);
}

int isic_conv_stem_fwd_stats_bf16(const uint16_t* in_nhwc4, const uint16_t* w_stem, uint16_t* out, int N, int Hin,
                                  int Win, int Hout, int Wout, double* stat_sum, double* stat_sumsq, int stat_slots,
                                  void* stream) {
  ISIC_CHECK_ARG(in_nhwc4 && w_stem && out && N > 0 && Hin > 0 && Win > 0);
  ISIC_CHECK_ARG((stat_sum == nullptr) == (stat_sumsq == nullptr));
  ISIC_CHECK_ARG(!stat_sum || stat_slots > 0);
  StemArgs a;
  a.stat_sum = stat_sum; a.stat_sumsq = stat_sumsq; a.stat_slots = stat_slots > 0 ? stat_slots : 1;
  a.in = in_nhwc4; a.w = w_stem; a.out = out; a.dy = nullptr; a.dw = nullptr;
  int rc = stem_args(a, N, Hin, Win, Hout, Wout);
  if (rc != ISIC_OK) return rc;
  const int grid = a.total_tiles < STEM_FWD_BLOCKS ? a.total_tiles : STEM_FWD_BLOCKS;
  hipLaunchKernelGGL(conv_stem_fwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), a);
  return isic_launch_status();
}

size_t isic_conv_stem_wgrad_workspace_bytes(void) { return (size_t)STEM_WGRAD_BLOCKS * STEM_DW_ELEMS * sizeof(float); }

int isic_conv_stem_wgrad_bf16(const uint16_t* in_nhwc4, const uint16_t* dy, float* dw, int N, int Hin, int Win,
                              int Hout, int Wout, void* workspace, size_t workspace_bytes, void* stream) {
  ISIC_CHECK_ARG(in_nhwc4 && dy && dw && workspace && N > 0 && Hin > 0 && Win > 0);
  if (workspace_bytes < isic_conv_stem_wgrad_workspace_bytes()) return ISIC_ERR_WORKSPACE;
  StemArgs a;
  a.stat_sum = nullptr; a.stat_sumsq = nullptr; a.stat_slots = 1;
  a.in = in_nhwc4; a.w = nullptr; a.out = nullptr; a.dy = dy; a.dw = reinterpret_cast<float*>(workspace);
  int rc = stem_args(a, N, Hin, Win, Hout, Wout);
  if (rc != ISIC_OK) return rc;
  const int grid = a.total_tiles < STEM_WGRAD_BLOCKS ? a.total_tiles : STEM_WGRAD_BLOCKS;
  hipLaunchKernelGGL(conv_stem_wgrad_kernel, dim3(grid), dim3(256), 0, as_stream(stream), a);
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(ceil_div(STEM_DW_ELEMS, 16)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float*>(workspace), dw, grid);
  return isic_launch_status();
}

int isic_conv_stem_wgrad_bn_pooled_bf16(const uint16_t* in_nhwc4, const uint16_t* y0, const uint8_t* argmax,
                                        const uint16_t* dy_pooled, const float* mean, const float* rstd,
                                        const float* gamma, const float* scale, const float* shift, const double* dgamma,
                                        const double* dbeta, float* dw, float* dgamma_f32, float* dbeta_f32, int N,
                                        int Hin, int Win, int Hout, int Wout, int Hp, int Wp, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  ISIC_CHECK_ARG(in_nhwc4 && y0 && argmax && dy_pooled && mean && rstd && gamma && scale && shift && dgamma && dbeta && dw);
  ISIC_CHECK_ARG(workspace != nullptr);
  if (workspace_bytes < isic_conv_stem_wgrad_workspace_bytes()) return ISIC_ERR_WORKSPACE;
  ISIC_CHECK_ARG(N > 0 && Hin > 0 && Win > 0 && (dgamma_f32 == nullptr) == (dbeta_f32 == nullptr));
  ISIC_CHECK_ARG(Hp == (Hout + 2 - 3) / 2 + 1 && Wp == (Wout + 2 - 3) / 2 + 1);
  StemBnArgs b;
  StemArgs& a = b.s;
  a.stat_sum = nullptr; a.stat_sumsq = nullptr; a.stat_slots = 1;
  a.in = in_nhwc4; a.w = nullptr; a.out = nullptr; a.dy = nullptr; a.dw = reinterpret_cast<float*>(workspace);
  int rc = stem_args(a, N, Hin, Win, Hout, Wout);
  if (rc != ISIC_OK) return rc;
  b.y0 = y0; b.argmax = argmax; b.gp = dy_pooled; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.scale = scale;
  b.shift = shift; b.dgamma = dgamma; b.dbeta = dbeta; b.dgamma_f32 = dgamma_f32; b.dbeta_f32 = dbeta_f32;
  b.Hp = Hp; b.Wp = Wp;
  const int grid = a.total_tiles < STEM_WGRAD_BLOCKS ? a.total_tiles : STEM_WGRAD_BLOCKS;
  hipLaunchKernelGGL(conv_stem_wgrad_bn_kernel, dim3(grid), dim3(256), 0, as_stream(stream), b);
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(ceil_div(STEM_DW_ELEMS, 16)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float*>(workspace), dw, grid);
  return isic_launch_status();
}

int isic_conv_stem_pack_bf16(const float* w_krsc, uint16_t* w_stem, void* stream) {
  ISIC_CHECK_ARG(w_krsc && w_stem);
  hipLaunchKernelGGL(stem_pack_kernel, dim3(ceil_div(64 * 224, 256)), dim3(256), 0, as_stream(stream), w_krsc, w_stem);
  return isic_launch_status();
}

}  // extern "C"
