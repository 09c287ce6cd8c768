// Row-panel GEMM for the node-wise Linears of the graph path:  C[M,N] = act(A[M,K] op(B) + bias) + beta * C (+ addend)  with a LONG M
// (all the nodes of a batch: 50,176 at BASELINE.json configs[3]), a SHORT reduction K <= 128 and N a multiple of 64 --
// GCNConv.lin forward and data gradient (128 -> 128), the attention heads' first Linear (128 -> 4 x 128).  Exact fp32 on
// v_mfma_f32_16x16x4_f32 (gfx950).  Reference: 05_train_gnns.py:82 (GCNConv), :126-131 (attention_layers), :184-185.
//
// At K = 128 such a product moves as many bytes as it multiplies (51 MB / 1.64 GFLOP for 128 -> 128: 8.5 us of HBM, 10.4 us of
// MFMA) and a tiled kernel spends a third of its time filling its operand ring per tile and on 256-row tiles that do not
// divide the rows over 256 CUs (196 tiles).  Here:
//   * the WHOLE weight slice (128 output columns x K) sits in LDS for the life of the block, transposed / permuted once;
//   * the unit of work is a 16-row panel x 64 columns: 12,544 / 6,272 units for 4 x 256 SIMDs (>= 94 % balanced), dealt
//     block-major so that a remainder spreads over the CUs;
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

template <int KC>                        // KC = K / 16 chunks of 16 k
__global__ __launch_bounds__(RP_WAVES * 64) void gemm_rowpanel_kernel(RpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wt[];         // [128 permuted columns][K + 4]
  const int K = KC * 16, PITCH = K + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int grid = gridDim.x, waves_all = grid * RP_WAVES;
  const int widx = wave * grid + blockIdx.x;                         // block-major deal: a remainder spreads over the CUs

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
    __syncthreads();

    const int units = a.panels * halves;
    f32x4 an[KC];                                                    // the next unit's A panel
    auto load_panel = [&](int u) {
      const int p = halves == 2 ? (u >> 1) : u;
      const int r = p * 16 + li;
      const bool ok = u < units && r < a.M;
      const float* src = a.A + (size_t)(ok ? r : 0) * a.lda + 4 * lq;
#pragma unroll
      for (int c = 0; c < KC; ++c) an[c] = ok ? *reinterpret_cast<const f32x4*>(src + 16 * c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    int u = widx;
    load_panel(u);
    for (; u < units; u += waves_all) {
      f32x4 af[KC];
#pragma unroll
      for (int c = 0; c < KC; ++c) af[c] = an[c];
      load_panel(u + waves_all);
      const int p = halves == 2 ? (u >> 1) : u, hf = halves == 2 ? (u & 1) : 0;
      const float* wb = wt + (hf * 64 + li) * PITCH + 4 * lq;        // tile t: + 16 t rows
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][x], bf[c & 1][t][x], acc[t], 0, 0, 0);
      }
      // acc[t][v] = C[16 p + 4 lq + v][n0 + 64 hf + 4 li + t]
      const int col = n0 + hf * 64 + 4 * li;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + col);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = p * 16 + 4 * lq + v;
        if (row >= a.M) continue;
        f32x4 o = {acc[0][v] + bv[0], acc[1][v] + bv[1], acc[2][v] + bv[2], acc[3][v] + bv[3]};
        if (a.act == ISIC_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        } else if (a.act == ISIC_ACT_TANH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = isic_tanhf(o[e]);
        }
        f32x4* cp = reinterpret_cast<f32x4*>(a.C + (size_t)row * a.ldc + col);
        if (a.beta != 0.f) o += a.beta * (*cp);
        if (a.addend) o += *reinterpret_cast<const f32x4*>(a.addend + (size_t)row * a.ldadd + col);
        *cp = o;
      }
    }
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

template <int KC>
int rp_launch(const RpArgs& a, hipStream_t stream) {
  const int lds = RP_SLICE * (KC * 16 + 4) * (int)sizeof(float);
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (lds > 48 * 1024 &&
      isic_once_per_device(once, [lds] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowpanel_kernel<KC>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  const int cus = isic_cu_count();
  const int units = a.panels * (a.N >= RP_SLICE ? 2 : 1);
  int grid = ceil_div(units, RP_WAVES);
  if (grid > cus) grid = cus;
  hipLaunchKernelGGL(gemm_rowpanel_kernel<KC>, dim3(grid), dim3(RP_WAVES * 64), lds, stream, a);
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
    case 1: return rp_launch<1>(a, stream);
    case 2: return rp_launch<2>(a, stream);
    case 3: return rp_launch<3>(a, stream);
    case 4: return rp_launch<4>(a, stream);
    case 5: return rp_launch<5>(a, stream);
    case 6: return rp_launch<6>(a, stream);
    case 7: return rp_launch<7>(a, stream);
    case 8: return rp_launch<8>(a, stream);
  }
  return ISIC_ERR_UNSUPPORTED;
}
