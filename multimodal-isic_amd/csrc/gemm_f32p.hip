// Persistent exact-fp32 GEMM on v_mfma_f32_16x16x4_f32 (gfx950) for the node-feature products of the graph path:
//   C[M,N] = act(op(A)[M,K] . op(B)[K,N] + bias[N]) + beta * C        (any transposition of A and B, C or C^T stored)
// i.e. the big nn.Linear layers of GraphMIL and their autograd backward at a batch of graphs
// (05_train_gnns.py:66,112,126-139,168-199: input_proj 50 176 x 768 -> 128, the 128-wide GCN / attention-head layers,
// dX = dY.W and dW = dY^T.X over all the nodes of a step) and of the MIL head (utils_g_mil.py:49-63).
//
// The 64 x 64 x 16 kernel of gemm_f32.hip stages through registers with two __syncthreads per 16 k and reaches 50-60 of
// the 157 TFLOP/s the fp32 matrix core offers.  This is the conv_pgemm / gemm_f16 machine instead:
//   * 1024 threads: waves 0-7 multiply (4 x 2 waves of 64 x 64 = 4 x 4 MFMA tiles, 64 accumulator VGPRs), waves 8-15 only
//     issue LDS-DMA (global_load_lds_dwordx4, 1 KB per instruction) into a three-stage ring of 48 KB K-tiles
//     (256 rows of A + 128 rows of B, 32 k each), one s_barrier per K-tile, counted vmcnt;
//   * a K-tile is 128 MFMAs per wave = 4096 cycles against 6 DMAs per staging wave: the matrix core is the only limit;
//   * BOTH operand layouts are staged without a transposing pass:
//       k contiguous (x[M,K], W[N,K]):  a stage row = 32 k of one matrix row, 16-byte chunks XOR-swizzled with row & 7 on
//           the SOURCE side; one ds_read_b128 per lane feeds four k-steps (lane group g supplies k = 16h + 4g + j);
//       k strided   (dY[K,M], X[K,N], W[K,N]): a stage row = one k of all 256 (128) matrix rows, chunks XOR-swizzled with
//           ((k >> 2) & 3) << 2 so that the four lane groups of a ds_read_b32 (k = 16h + 4g + j) hit 64 different banks;
//     both use the same k of a lane at the same MFMA step, so the two layouts mix freely (dX = dY.W is contiguous x strided);
//   * persistent blocks walk a contiguous range of 256-row tiles of one 128-column slice; SPLIT-K for a long reduction
//     into a small output (weight gradients: K = all nodes of the step) writes per-split partial tiles into the caller's
//     workspace and a second kernel adds them IN SPLIT ORDER and applies bias / activation / beta: no atomics, the result
//     is bit-reproducible (the old kernel's split-K used fp32 atomics in arrival order);
//   * register-only epilogue: operand roles are swapped in the MFMA so that a lane ends with four consecutive columns of
//     one row (16-byte stores), or of one column of C^T when the caller wants the transpose (dW computed as X^T.dY so
//     that the 768-wide side fills the 256-row tile).
// Shapes the DMA path cannot take (a dimension that is not a multiple of 4 floats, unaligned pointers) and small products
// stay on gemm_f32.hip's kernel: isic_gemm_f32_ws decides.
#include "common.h"

namespace {

constexpr int PM = 256, PN = 128, PK = 32;
constexpr int P_A = PM * PK * 4, P_B = PN * PK * 4, P_STAGE = P_A + P_B;     // 32 KB + 16 KB
constexpr int P_NST = 3;
constexpr int P_PER_IT = 6;                    // DMAs per staging wave and K-tile (4 of A, 2 of B)
constexpr int P_LDS = P_NST * P_STAGE + 1024;  // ring | DMA scratch

struct G32Args {
  const float* A;          // AK: [M][lda] (k contiguous)   else [K][lda] (m contiguous)
  const float* B;          // BK: [N][ldb] (k contiguous)   else [K][ldb] (n contiguous)
  float* C;                // [M][ldc], or [N][ldc] when transC
  float* partial;          // split-K: [splits][M][N] raw sums (N % 4 == 0), else null
  const float* bias;       // [N] or null
  const int* arow;         // or null: row r of the kernel's A operand (m for AK, k otherwise) is stored row arow[r] of A
  int M, N, K, lda, ldb, ldc;
  int act, transC, vecC;   // vecC: 16-byte stores into C are legal (ldc % 4 == 0, aligned, N % 4 == 0)
  float beta;
  int Ktiles, kt_per_split, splits;
  int mtiles, tiles_per_block;
};

__device__ __attribute__((aligned(256))) unsigned char g_g32_zero_page[256];

__device__ __forceinline__ void g32_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// Row indices of a gathered operand are read with SCALAR loads (lgkmcnt): the staging waves count vmcnt by hand, a vector
// load among their LDS-DMAs would break the count.  The address is wave-uniform by construction.
typedef int g32_i8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ g32_i8 g32_sload8(const int* base_uniform) {
  const unsigned long long q = (unsigned long long)base_uniform;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)q), hi = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32));
  const unsigned long long qs = ((unsigned long long)hi << 32) | lo;
  g32_i8 v;
  asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(qs) : "memory");
  return v;
}
__device__ __forceinline__ int g32_sload1(const int* p_uniform) {
  const unsigned long long q = (unsigned long long)p_uniform;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)q), hi = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32));
  const unsigned long long qs = ((unsigned long long)hi << 32) | lo;
  int v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(qs) : "memory");
  return v;
}

__device__ __forceinline__ float g32_act(float v, int act) {
  return act == ISIC_ACT_RELU ? fmaxf(v, 0.f) : (act == ISIC_ACT_TANH ? isic_tanhf(v) : v);
}

template <bool AK, bool BK>
__global__ __launch_bounds__(1024) void gemm_f32p_kernel(G32Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  constexpr int off_scr = P_NST * P_STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.y * PN;
  const int t_begin = blockIdx.x * a.tiles_per_block;
  const int ntl = min(a.mtiles - t_begin, a.tiles_per_block);
  if (ntl <= 0) return;                                    // whole block: no barrier has been reached yet
  const int kt0 = blockIdx.z * a.kt_per_split;
  const int KT = min(a.Ktiles - kt0, a.kt_per_split);      // >= 1 (host: splits = ceil(Ktiles / kt_per_split))
  const int total_it = ntl * KT;

  if (wave >= 8) {
    // =================================================================== staging waves
    const int sw = wave - 8;
    const unsigned char* zp = g_g32_zero_page + (lane & 7) * 16;
    const unsigned scr = lds0 + off_scr;
    // ---- A: k contiguous: four instructions, rows 8 (sw + 8 i) + (lane >> 3), global chunk (lane & 7) ^ (lane >> 3)
    //         k strided:    four instructions, k rows sw + 8 i, lane l lands at chunk l and fetches chunk l ^ key(k)
    const int r8 = lane >> 3;
    const int gch = (lane & 7) ^ r8;
    const float* a_ptr[4];
    bool a_ok[4];
    auto tile_rows = [&](int tl) {
      const int m0 = (t_begin + tl) * PM;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (AK) {
          const int mb = m0 + 8 * (sw + 8 * i);            // wave-uniform; rows mb .. mb + 7, one per r8
          const int m = mb + r8;
          a_ok[i] = m < a.M;
          int src = a_ok[i] ? m : 0;
          if (a.arow && mb < a.M) {                        // gathered rows (the index array is padded to a multiple of 8)
            const g32_i8 v = g32_sload8(a.arow + mb);
            int pick = v[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) pick = r8 == q ? v[q] : pick;
            src = a_ok[i] ? pick : 0;
          }
          a_ptr[i] = a.A + (size_t)src * a.lda + gch * 4;
        } else {
          const int k = sw + 8 * i;                        // k row inside the K-tile
          const int m = m0 + 4 * (lane ^ (((k >> 2) & 3) << 2));
          a_ok[i] = m < a.M;                               // M % 4 == 0 (host): the whole chunk is inside
          a_ptr[i] = a.A + (size_t)k * a.lda + (a_ok[i] ? m : 0);
        }
      }
    };
    // ---- B (the same 128 columns for every tile of the block)
    const float* b_ptr[2];
    bool b_ok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (BK) {
        const int n = n0 + 16 * sw + 8 * t + r8;
        b_ok[t] = n < a.N;
        b_ptr[t] = a.B + (size_t)(b_ok[t] ? n : 0) * a.ldb + gch * 4;
      } else {
        const int k = 2 * (2 * sw + t) + (lane >> 5);      // two k rows of 512 B per instruction
        const int n = n0 + 4 * ((lane & 31) ^ (((k >> 2) & 3) << 2));
        b_ok[t] = n < a.N;                                 // N % 4 == 0 (host)
        b_ptr[t] = a.B + (size_t)k * a.ldb + (b_ok[t] ? n : 0);
      }
    }
    auto issue = [&](int kt, int stage, bool live) {
      const unsigned sbase = lds0 + stage * P_STAGE;
      const int k0 = (kt0 + kt) * PK;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const void* src;
        if (AK) {
          // a 16-byte chunk holds k0 + 4 gch .. + 3: K % 4 == 0 (host), so it is inside or outside as a whole
          src = (live && a_ok[i] && k0 + gch * 4 < a.K) ? (const void*)(a_ptr[i] + k0) : (const void*)zp;
        } else {
          const bool in = live && k0 + sw + 8 * i < a.K;   // wave-uniform
          size_t roff = (size_t)k0 * a.lda;                // a_ptr[i] already holds row sw + 8 i
          if (a.arow && in)                                // gathered k rows: stored row arow[k] instead of row k
            roff = (size_t)g32_sload1(a.arow + k0 + sw + 8 * i) * a.lda - (size_t)(sw + 8 * i) * a.lda;
          src = (in && a_ok[i]) ? (const void*)(a_ptr[i] + roff) : (const void*)zp;
        }
        g32_glds16(src, live ? sbase + (unsigned)((sw + 8 * i) * 1024) : scr);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const void* src;
        if (BK) {
          src = (live && b_ok[t] && k0 + gch * 4 < a.K) ? (const void*)(b_ptr[t] + k0) : (const void*)zp;
        } else {
          const int k = 2 * (2 * sw + t) + (lane >> 5);
          src = (live && b_ok[t] && k0 + k < a.K) ? (const void*)(b_ptr[t] + (size_t)k0 * a.ldb) : (const void*)zp;
        }
        g32_glds16(src, live ? sbase + P_A + (unsigned)((2 * sw + t) * 1024) : scr);
      }
    };
    int itile = 0, ikt = 0;
    tile_rows(0);
    auto advance = [&]() {
      if (++ikt == KT) { ikt = 0; ++itile; if (itile < ntl) tile_rows(itile); }
    };
    issue(0, 0, true); advance();
    issue(ikt, 1, total_it > 1); advance();
    for (int it = 0; it < total_it; ++it) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_PER_IT) : "memory");      // K-tile `it` has landed
      __builtin_amdgcn_s_barrier();
      issue(ikt, (it + 2) % P_NST, it + 2 < total_it);
      advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA may outlive the block's LDS allocation
  } else {
    // ===================================================================== MFMA waves
    const int fr = lane & 15, fg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // k contiguous: 128-byte rows, chunk (4h + fg) ^ (row & 7); rows 16 apart share the key: tile i = + 2048 i
    const unsigned ak_off = (unsigned)((wm * 64 + fr) * 128 + ((fg ^ (fr & 7)) << 4));
    const unsigned bk_off = (unsigned)(P_A + (wn * 64 + fr) * 128 + ((fg ^ (fr & 7)) << 4));
    // k strided: row k = 16h + 4fg + j of 1024 B (A) / 512 B (B); column c lives in chunk (c >> 2) ^ (fg << 2): with
    // c = w*64 + i*16 + fr that is chunk w*16 + ((i ^ fg) << 2) + (fr >> 2)
    const unsigned as_off = (unsigned)(4 * fg * 1024 + ((wm * 16 + (fr >> 2)) << 4) + (fr & 3) * 4);
    const unsigned bs_off = (unsigned)(P_A + 4 * fg * 512 + ((wn * 16 + (fr >> 2)) << 4) + (fr & 3) * 4);

    int it = 0;
    for (int tl = 0; tl < ntl; ++tl) {
      const int m0 = (t_begin + tl) * PM;
      f32x4 acc[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
      for (int kt = 0; kt < KT; ++kt, ++it) {
        __builtin_amdgcn_s_barrier();
        const unsigned char* st = smem + (it % P_NST) * P_STAGE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 af[4], bf[4];                              // [tile][j]: the operand of k-step j
          if (AK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f32x4*>(st + ((ak_off ^ (unsigned)(h << 6)) + i * 2048));
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                af[i][j] = *reinterpret_cast<const float*>(st + as_off + (16 * h + j) * 1024 + (((i ^ fg) << 2) << 4));
          }
          if (BK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bf[i] = *reinterpret_cast<const f32x4*>(st + ((bk_off ^ (unsigned)(h << 6)) + i * 2048));
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                bf[i][j] = *reinterpret_cast<const float*>(st + bs_off + (16 * h + j) * 512 + (((i ^ fg) << 2) << 4));
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int jt = 0; jt < 4; ++jt)
                acc[i][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[jt][j], af[i][j], acc[i][jt], 0, 0, 0);
        }
      }

      // ---- register-only epilogue: lane (fg, fr) holds columns n0 + wn*64 + jt*16 + 4fg + {0..3} of row
      //      m0 + wm*64 + i*16 + fr for the 4 x 4 tiles (i, jt)
      // (32-bit element offsets against the scalar base pointers: 64-bit per-lane pointers hoisted out of the tile loop
      //  were spilled to scratch; the host guarantees M * ldc, N * ldc and splits * M * N < 2^31)
      int m0v = m0;
      asm volatile("" : "+v"(m0v));                        // keeps the address arithmetic inside the epilogue
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0v + wm * 64 + i * 16 + fr;
        if (m >= a.M) continue;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const int n = n0 + wn * 64 + jt * 16 + 4 * fg;
          if (n >= a.N) continue;
          f32x4 c = acc[i][jt];
          if (a.partial) {                                 // split-K: raw sums, [split][M][N], N % 4 == 0
            const unsigned off = (unsigned)((blockIdx.z * a.M + m) * a.N + n);
            *reinterpret_cast<f32x4*>(a.partial + off) = c;
            continue;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) c[e] = g32_act(c[e] + ((a.bias && n + e < a.N) ? a.bias[n + e] : 0.f), a.act);
          if (a.transC) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < a.N) {
                float* p = a.C + (unsigned)((n + e) * a.ldc + m);
                *p = a.beta != 0.f ? c[e] + a.beta * (*p) : c[e];
              }
          } else if (a.vecC) {
            f32x4* p = reinterpret_cast<f32x4*>(a.C + (unsigned)(m * a.ldc + n));
            if (a.beta != 0.f) { const f32x4 o = *p; c += a.beta * o; }
            *p = c;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < a.N) {
                float* p = a.C + (unsigned)(m * a.ldc + n + e);
                *p = a.beta != 0.f ? c[e] + a.beta * (*p) : c[e];
              }
          }
        }
      }
    }
  }   // MFMA waves
}

// C = act(sum over the splits, in a FIXED order, + bias) + beta * C.  partial: [splits][M][N].  Eight lanes per group of 4
// columns: lane g adds splits g, g + 8, ... (16-byte loads), a fixed xor tree joins them -- one thread per element group
// walked up to ~200 dependent loads (29 us for a 128 x 768 output and 85 splits).
__global__ __launch_bounds__(256) void gemm_f32p_reduce_kernel(G32Args a) {
  const int nv = a.N >> 2;
  const int g = threadIdx.x & 7;
  const int64_t idx = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool live = idx < (int64_t)a.M * nv;
  const int m = live ? (int)(idx / nv) : 0, n = live ? (int)(idx - (int64_t)m * nv) * 4 : 0;
  const size_t stride = (size_t)a.M * a.N;
  const float* p = a.partial + (size_t)m * a.N + n;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (live)
    for (int z = g; z < a.splits; z += 8) s += *reinterpret_cast<const f32x4*>(p + (size_t)z * stride);
#pragma unroll
  for (int o = 4; o > 0; o >>= 1)
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += __shfl_xor(s[e], o, 8);
  if (!live || g != 0) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float v = g32_act(s[e] + (a.bias ? a.bias[n + e] : 0.f), a.act);
    float* c = a.transC ? a.C + (size_t)(n + e) * a.ldc + m : a.C + (size_t)m * a.ldc + n + e;
    *c = a.beta != 0.f ? v + a.beta * (*c) : v;
  }
}

template <bool AK, bool BK>
int launch_g32(const G32Args& a, dim3 grid, hipStream_t stream) {
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (isic_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32p_kernel<AK, BK>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_f32p_kernel<AK, BK>), grid, dim3(1024), P_LDS, stream, a);
  return ISIC_OK;
}

struct G32Plan {
  bool ok, swap;           // swap: compute C^T = op(B)^T . op(A)^T so that the wide side fills the 256-row tile
  int splits, kt_per_split, mtiles, nslices, tiles_per_block, gx;
  size_t partial_bytes;
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Plan of the persistent kernel for C[M,N] (+)= op(A).op(B); `have_ws`: split-K partials can be parked in a workspace.
// Used by isic_gemm_f32_ws (gemm_f32.hip) and isic_gemm_f32_workspace_bytes.
static G32Plan g32_plan(int transA, int transB, int M, int N, int K) {
  G32Plan p{};
  // worth it only when the product fills the chip: ~2 x 10^8 FLOP and at least 128 k
  p.ok = K >= 64 && (double)M * N * K >= 1.0e8 && K % 4 == 0 && M % 4 == 0 && N % 4 == 0 &&
         (int64_t)M * N < (1LL << 31);                     // 32-bit element offsets
  if (!p.ok) return p;
  // roles: rows of the 256-row tile = the larger of M, N (a 128-row output would leave half of every tile empty)
  p.swap = N > M && M <= 128;
  const int Mr = p.swap ? N : M, Nr = p.swap ? M : N;
  p.mtiles = ceil_div(Mr, PM);
  p.nslices = ceil_div(Nr, PN);
  const int Ktiles = ceil_div(K, PK);
  // measured against the 64 x 64 x 16 kernel at 50 176 rows (tools/gemm_bench.py, profiles/r03_gemm_bench.txt): the 256-row
  // tile loses when the output has fewer than 192 rows (half of every tile is empty: 128 x 128 x 50 176 ran 88 vs 51 us) and
  // when a short reduction (K = 128: 4 K-tiles per tile) meets several column slices (196 x 4 work items quantise badly
  // over 256 CUs: 115 vs 108 us); it wins 1.15-1.55x everywhere else
  if (Mr < 192 || (Ktiles < 8 && p.nslices > 1)) { p.ok = false; return p; }
  const int cus = isic_cu_count();
  const int tiles = p.mtiles * p.nslices;
  p.splits = 1;
  if (tiles * 2 <= cus && Ktiles >= 16) {                  // long reduction into a small output: split K over the idle CUs
    int want = cus / tiles;
    if (want > Ktiles / 8) want = Ktiles / 8;              // >= 8 K-tiles (256 k) per split
    if (want > 1) p.splits = want;
  }
  p.kt_per_split = ceil_div(Ktiles, p.splits);
  p.splits = ceil_div(Ktiles, p.kt_per_split);
  if ((int64_t)M * N * p.splits >= (1LL << 31)) { p.ok = false; return p; }      // 32-bit offsets into the partials
  if (tiles * p.splits < 96) { p.ok = false; return p; }   // too few 256 x 128 work items for 256 CUs: the 64 x 64 kernel fills the chip better
  int groups = cus / (p.nslices * p.splits);
  if (groups < 1) groups = 1;
  p.tiles_per_block = ceil_div(p.mtiles, groups);
  p.gx = ceil_div(p.mtiles, p.tiles_per_block);
  p.partial_bytes = p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
  return p;
}

size_t isic_gemm_f32p_workspace_bytes(int transA, int transB, int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const G32Plan p = g32_plan(transA, transB, M, N, K);
  return p.ok ? p.partial_bytes : 0;
}

// Returns ISIC_ERR_UNSUPPORTED when the shape / alignment is not for this kernel (the caller falls back to the 64 x 64
// kernel), ISIC_OK after launching.
int isic_gemm_f32p_rows_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const int* a_rows,
                               const float* B, int ldb, const int* b_rows, float* C, int ldc, const float* bias, int act,
                               float beta, void* workspace, size_t workspace_bytes, hipStream_t stream);

int isic_gemm_f32p_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          float* C, int ldc, const float* bias, int act, float beta, void* workspace,
                          size_t workspace_bytes, hipStream_t stream) {
  return isic_gemm_f32p_rows_launch(transA, transB, M, N, K, A, lda, nullptr, B, ldb, nullptr, C, ldc, bias, act, beta, workspace,
                                    workspace_bytes, stream);
}

// ... with a row index on an operand: stored row r of A (of B) is taken from A[a_rows[r]] (B[b_rows[r]]): a batch of graphs is
// multiplied straight out of the resident record store, without materialising the gathered rows.  Only for the operand that
// ends up as the kernel's row-tile operand (x in x W^T; X in dY^T X with <= 128 output rows); ISIC_ERR_UNSUPPORTED otherwise.
int isic_gemm_f32p_rows_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const int* a_rows,
                               const float* B, int ldb, const int* b_rows, float* C, int ldc, const float* bias, int act,
                               float beta, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  G32Plan p = g32_plan(transA, transB, M, N, K);
  if (!p.ok || lda % 4 != 0 || ldb % 4 != 0 || !aligned16(A) || !aligned16(B)) return ISIC_ERR_UNSUPPORTED;
  if ((int64_t)(M > N ? M : N) * ldc >= (1LL << 31)) return ISIC_ERR_UNSUPPORTED;      // 32-bit element offsets into C
  if (p.splits > 1 && (workspace == nullptr || workspace_bytes < p.partial_bytes || !aligned16(workspace))) {
    // no room for the partials: one split (still correct, just fewer busy CUs)
    p.splits = 1; p.kt_per_split = ceil_div(K, PK);
    int groups = isic_cu_count() / p.nslices;
    if (groups < 1) groups = 1;
    p.tiles_per_block = ceil_div(p.mtiles, groups);
    p.gx = ceil_div(p.mtiles, p.tiles_per_block);
  }
  if ((!p.swap && b_rows) || (p.swap && a_rows)) return ISIC_ERR_UNSUPPORTED;       // only the row-tile operand gathers
  G32Args a;
  a.arow = p.swap ? b_rows : a_rows;
  a.bias = bias; a.act = act; a.beta = beta; a.C = C; a.ldc = ldc;
  a.K = K; a.Ktiles = ceil_div(K, PK); a.kt_per_split = p.kt_per_split; a.splits = p.splits;
  a.mtiles = p.mtiles; a.tiles_per_block = p.tiles_per_block;
  a.partial = p.splits > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
  bool AK, BK;
  if (!p.swap) {
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.M = M; a.N = N; a.transC = 0;
    AK = !transA;                  // A stored [M][K]: k contiguous
    BK = transB != 0;              // B stored [N][K]: k contiguous
  } else {
    // C^T[N,M] = op(B)^T[N,K] . op(A)^T[K,M]: the kernel's "A" is the caller's B with rows n, its "B" the caller's A
    a.A = B; a.lda = ldb; a.B = A; a.ldb = lda; a.M = N; a.N = M; a.transC = 1;
    AK = transB != 0;              // caller's B stored [N][K]: k contiguous for rows n
    BK = !transA;                  // caller's A stored [M][K]: k contiguous for rows m
  }
  // the bias belongs to the caller's columns n: after a swap those are the kernel's ROWS -- only the reduce kernel and
  // the un-swapped epilogue index it by column, so a swapped product must not carry one
  if (p.swap && bias) return ISIC_ERR_UNSUPPORTED;
  a.vecC = (!a.transC && ldc % 4 == 0 && aligned16(C) && a.N % 4 == 0) ? 1 : 0;
  const dim3 grid(p.gx, p.nslices, p.splits);
  int rc;
  if (AK && BK) rc = launch_g32<true, true>(a, grid, stream);
  else if (AK && !BK) rc = launch_g32<true, false>(a, grid, stream);
  else if (!AK && BK) rc = launch_g32<false, true>(a, grid, stream);
  else rc = launch_g32<false, false>(a, grid, stream);
  if (rc != ISIC_OK) return rc;
  if (p.splits > 1) {
    const int64_t items = (int64_t)a.M * (a.N >> 2);
    hipLaunchKernelGGL(gemm_f32p_reduce_kernel, dim3((unsigned)ceil_div64(items, 32)), dim3(256), 0, stream, a);
  }
  return isic_launch_status();
}
