// Row-panel GEMM for the node-wise Linears of the graph path:  C[M,N] = act(A[M,K] op(B) + bias) + beta * C (+ addend)  with a LONG M
// (all the nodes of a batch: 50,176 at BASELINE.json configs[3]), a SHORT reduction K <= 128 and N a multiple of 64 --
// GCNConv.lin forward and data gradient (128 -> 128), the attention heads' first Linear (128 -> 4 x 128).  Exact fp32 on
// v_mfma_f32_16x16x4_f32 (gfx950).  Reference: 05_train_gnns.py:82 (GCNConv), :126-131 (attention_layers), :184-185.
//
// At K = 128 such a product moves as many bytes as it multiplies (51 MB / 1.64 GFLOP for 128 -> 128: 8.5 us of HBM, 10.4 us of
// MFMA) and a tiled kernel spends a third of its time filling its operand ring per tile and on 256-row tiles that do not
// divide the rows over 256 CUs (196 tiles).  Here:
//   * the WHOLE weight slice (128 output columns x K) sits in LDS for the life of the block, transposed / permuted once;
//   * the unit of work is a 16-row panel x 64 columns, panels dealt round-robin to the CUs and a CU's units round-robin to its
//     8 waves: 24.5 units per CU for 4 SIMDs at 50,176 rows (87 % balanced), both halves of a panel on the same CU;
//   * A goes global -> registers, no LDS: lane (i = l & 15, q = l >> 4) loads float4 A[r0 + i][16 c + 4 q ..+3], which is
//     the A operand of four consecutive MFMA k-steps (any order of k is a valid summation order as long as B follows it);
//     the next unit's panel is in flight while this one multiplies;
//   * column tile t of the 64 holds columns 4 j + t (j = l & 15), so that a lane ends up with FOUR CONSECUTIVE columns of
//     four rows: 16-byte stores, 256 contiguous bytes per row; bias / ReLU / tanh / beta in the same registers.
#include "common.h"

namespace {

constexpr int RP_SLICE = 128;            // output columns whose weights are resident at a time
constexpr int RP_WAVES = 8;

struct RpArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* addend;                   // [M][ldadd] added after the activation (a residual / second gradient path), or NULL
  int M, N, K, lda, ldb, ldc, ldadd, transB, act;
  float beta;
  int panels;                            // ceil(M / 16)
};

// 16-byte global load OUTSIDE hipcc's vmcnt bookkeeping: the compiler's s_waitcnt insertion is conservative across the
// back-edge of the unit loop -- it waited for the panel it had just requested (vmcnt(0) in front of the first MFMA) and for
// every store's round trip before the next one (tests with the loads / stores compiled out: 14 us each of an 84 us launch,
// none of it overlapped).  The waves count by hand instead: rp_wait<N>() = "all but the N youngest requests have landed".
__device__ __forceinline__ f32x4 rp_gload16(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int KC, bool ADD>              // KC = K / 16 chunks of 16 k; ADD: an addend[M][ldadd] joins in the epilogue
__global__ __launch_bounds__(RP_WAVES * 64) void gemm_rowpanel_kernel(RpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wt[];         // [128 permuted columns][K + 4]
  const int K = KC * 16, PITCH = K + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int grid = gridDim.x;

  for (int n0 = 0; n0 < a.N; n0 += RP_SLICE) {
    const int width = min(RP_SLICE, a.N - n0);                       // 128 or 64
    const int halves = width >> 6;
    // ---- the slice's weights into LDS: row rho(n) = 64 (n / 64) + 16 (n & 3) + ((n & 63) / 4) holds column n0 + n, k contiguous
    if (n0 > 0) __syncthreads();
    if (a.transB) {                      // B = W[N][K]
      for (int e = tid; e < width * (K / 4); e += RP_WAVES * 64) {
        const int n = e / (K / 4), k4 = (e - n * (K / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.B + (size_t)(n0 + n) * a.ldb + k4);
        const int rho = (n & 64) + 16 * (n & 3) + ((n & 63) >> 2);
        *reinterpret_cast<f32x4*>(&wt[rho * PITCH + k4]) = v;
      }
    } else {                             // B = W[K][N]
      for (int e = tid; e < K * (width / 4); e += RP_WAVES * 64) {
        const int k = e / (width / 4), n4 = (e - k * (width / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.B + (size_t)k * a.ldb + n0 + n4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int n = n4 + c;
          wt[((n & 64) + 16 * (n & 3) + ((n & 63) >> 2)) * PITCH + k] = v[c];
        }
      }
    }
    // the bias of the wave's two possible column groups (compiler-tracked loads: waited for here, once per slice)
    f32x4 bias2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (a.bias) {
      bias2[0] = *reinterpret_cast<const f32x4*>(a.bias + n0 + 4 * li);
      if (halves == 2) bias2[1] = *reinterpret_cast<const f32x4*>(a.bias + n0 + 64 + 4 * li);
    }
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the hand count below starts from zero

    // Panels are dealt to the blocks round-robin (panel p -> block p % grid); a block walks the units of ITS panels -- the two
    // 64-column halves of a panel are consecutive units, i.e. two neighbouring waves of the same CU in the same round.
    const int my_panels = (a.panels - (int)blockIdx.x + grid - 1) / grid;          // panels blockIdx.x, + grid, ...
    const int units = my_panels * halves;
    f32x4 an[KC], dn[4];                                             // the next unit's A panel (and addend rows)
#pragma unroll
    for (int v = 0; v < 4; ++v) dn[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // requests of unit j: KC panel loads (+ 4 addend loads): rows past M are clamped (loaded, never stored)
    auto request = [&](int j) {
      const int p = (int)blockIdx.x + (halves == 2 ? (j >> 1) : j) * grid;
      const int r = min(p * 16 + li, a.M - 1);
      const float* src = a.A + (size_t)r * a.lda + 4 * lq;
#pragma unroll
      for (int c = 0; c < KC; ++c) an[c] = rp_gload16(src + 16 * c);
      if (ADD) {
        const int col = n0 + (halves == 2 ? (j & 1) : 0) * 64 + 4 * li;
#pragma unroll
        for (int v = 0; v < 4; ++v) dn[v] = rp_gload16(a.addend + (size_t)min(p * 16 + 4 * lq + v, a.M - 1) * a.ldadd + col);
      }
    };
    // A requested register may only be touched after its wait -- and the compiler moves loop-carried values around at the END
    // of a loop body (phi copies).  So the wait for the NEXT unit's requests closes the iteration that issued them (they had
    // the whole multiply phase to land), and what the loop carries is the landed value.
    auto landed = [&](int allow4) {
      if (allow4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but the four stores issued after the requests
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // (volatile asm keeps its order; the empty ones make every later use of a requested register depend on the wait)
#pragma unroll
      for (int c = 0; c < KC; ++c) asm volatile("" : "+v"(an[c]));
      if (ADD) {
#pragma unroll
        for (int v = 0; v < 4; ++v) asm volatile("" : "+v"(dn[v]));
      }
    };
    int u = wave;
    if (u < units) { request(u); landed(0); }
    for (; u < units; u += RP_WAVES) {
      f32x4 af[KC], ad[4];
#pragma unroll
      for (int c = 0; c < KC; ++c) af[c] = an[c];
#pragma unroll
      for (int v = 0; v < 4; ++v) ad[v] = dn[v];
      // (no branch around the requests -- a join would make the compiler copy the requested registers at once, before they
      //  land: the last unit asks for itself again)
      request(u + RP_WAVES < units ? u + RP_WAVES : u);
      const int p = (int)blockIdx.x + (halves == 2 ? (u >> 1) : u) * grid, hf = halves == 2 ? (u & 1) : 0;
      const float* wb = wt + (hf * 64 + li) * PITCH + 4 * lq;        // tile t: + 16 t rows
      // TWO accumulators per column tile (even / odd k-steps), joined at the end: a v_mfma_f32_16x16x4_f32 can follow one that
      // wrote the same accumulator only ~6 issue slots later (tests/probes/probe_mfma_f32.hip: 4 accumulators in the ring
      // sustain 0.65 of the rate 16 do), and a unit has only four column tiles
      f32x4 acc[4], acc2[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      f32x4 bf[2][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bf[0][t] = *reinterpret_cast<const f32x4*>(wb + t * 16 * PITCH);
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        if (c + 1 < KC) {
#pragma unroll
          for (int t = 0; t < 4; ++t) bf[(c + 1) & 1][t] = *reinterpret_cast<const f32x4*>(wb + t * 16 * PITCH + 16 * (c + 1));
        }
#pragma unroll
        for (int x = 0; x < 4; x += 2) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][x], bf[c & 1][t][x], acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][x + 1], bf[c & 1][t][x + 1], acc2[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] += acc2[t];
      // acc[t][v] = C[16 p + 4 lq + v][n0 + 64 hf + 4 li + t]
      const int col = n0 + hf * 64 + 4 * li;
      const f32x4 bv = hf ? bias2[1] : bias2[0];
      f32x4 o[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        o[v] = (f32x4){acc[0][v] + bv[0], acc[1][v] + bv[1], acc[2][v] + bv[2], acc[3][v] + bv[3]};
        if (a.act == ISIC_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[v][e] = fmaxf(o[v][e], 0.f);
        } else if (a.act == ISIC_ACT_TANH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[v][e] = isic_tanhf(o[v][e]);
        }
        if (ADD) o[v] += ad[v];
      }
      float* crow = a.C + (size_t)(p * 16 + 4 * lq) * a.ldc + col;
      if (p * 16 + 16 <= a.M && a.beta == 0.f) {                     // wave-uniform: a full panel, four stores, nothing to wait for
#pragma unroll
        for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(crow + (size_t)v * a.ldc) = o[v];
        landed(1);
      } else {                                                       // the last panel / beta: tracked loads, predicated stores
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (p * 16 + 4 * lq + v < a.M) {
            f32x4* cp = reinterpret_cast<f32x4*>(crow + (size_t)v * a.ldc);
            if (a.beta != 0.f) o[v] += a.beta * (*cp);
            *cp = o[v];
          }
        }
        landed(0);                                                   // unknown number of stores: drain everything
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (requests never outlive the slice)
  }
}

bool rp_ok(int transA, int M, int N, int K, int lda, int ldb, int ldc, const void* A, const void* B, const void* C,
           const void* bias, const void* addend, int ldadd) {
  if (transA || M < 4096 || K > 128 || K < 16 || K % 16 != 0 || N % 64 != 0 || N < 64) return false;
  if ((lda & 3) || (ldb & 3) || (ldc & 3) || (addend && (ldadd & 3))) return false;
  const uintptr_t bits = reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C) |
                         reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(addend);
  return (bits & 15) == 0;
}

template <int KC, bool ADD>
int rp_launch(const RpArgs& a, hipStream_t stream) {
  const int lds = RP_SLICE * (KC * 16 + 4) * (int)sizeof(float);
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (lds > 48 * 1024 &&
      isic_once_per_device(once, [lds] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowpanel_kernel<KC, ADD>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  const int cus = isic_cu_count();
  int grid = ceil_div(a.panels * (a.N >= RP_SLICE ? 2 : 1), RP_WAVES);
  if (grid > cus) grid = cus;
  hipLaunchKernelGGL((gemm_rowpanel_kernel<KC, ADD>), dim3(grid), dim3(RP_WAVES * 64), lds, stream, a);
  return isic_launch_status();
}

}  // namespace

// ISIC_ERR_UNSUPPORTED: not a shape for this kernel (the caller falls through to the other kernels)
int isic_gemm_f32r_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          float* C, int ldc, const float* bias, int act, float beta, const float* addend, int ldadd,
                          hipStream_t stream) {
  if (!rp_ok(transA, M, N, K, lda, ldb, ldc, A, B, C, bias, addend, ldadd)) return ISIC_ERR_UNSUPPORTED;
  RpArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.addend = addend; a.ldadd = ldadd;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.transB = transB; a.act = act; a.beta = beta;
  a.panels = ceil_div(M, 16);
  switch (K / 16) {
    case 1: return addend ? rp_launch<1, true>(a, stream) : rp_launch<1, false>(a, stream);
    case 2: return addend ? rp_launch<2, true>(a, stream) : rp_launch<2, false>(a, stream);
    case 3: return addend ? rp_launch<3, true>(a, stream) : rp_launch<3, false>(a, stream);
    case 4: return addend ? rp_launch<4, true>(a, stream) : rp_launch<4, false>(a, stream);
    case 5: return addend ? rp_launch<5, true>(a, stream) : rp_launch<5, false>(a, stream);
    case 6: return addend ? rp_launch<6, true>(a, stream) : rp_launch<6, false>(a, stream);
    case 7: return addend ? rp_launch<7, true>(a, stream) : rp_launch<7, false>(a, stream);
    case 8: return addend ? rp_launch<8, true>(a, stream) : rp_launch<8, false>(a, stream);
  }
  return ISIC_ERR_UNSUPPORTED;
}
