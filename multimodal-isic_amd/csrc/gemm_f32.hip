// Exact-fp32 dense GEMM on v_mfma_f32_16x16x4_f32 (gfx950).
//   C[M,N] = act(op(A)[M,K] * op(B)[K,N] + bias[N]) + beta * C
// Replaces the nn.Linear layers (and their autograd backward) of the reference's
// MIL head (utils_g_mil.py:49-63), GraphMIL (05_train_gnns.py:66,112,126-139)
// and radiomic MLP / fusion head (model.py:74-83,138-143).
//
// Tiling: 256 threads = 4 waves (2x2), block tile 64x64, BK = 16; each wave owns
// a 32x32 sub-tile = 2x2 MFMA tiles.  Both operands are staged in LDS as
// [row][k] with k contiguous, so one ds_read_b128 feeds four MFMA k-steps: within
// a 16-deep chunk lane-group g (= lane>>4) supplies k = 4g + j at step j for A
// and for B alike (any permutation of k is a valid summation order as long as
// both operands use it).
#include "common.h"
#include "../../include/isic_hip_test.h"

// persistent LDS-DMA kernel for the large products (gemm_f32p.hip); ISIC_ERR_UNSUPPORTED = "not for this shape"
size_t isic_gemm_f32p_workspace_bytes(int transA, int transB, int M, int N, int K);
int isic_gemm_f32p_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          float* C, int ldc, const float* bias, int act, float beta, void* workspace,
                          size_t workspace_bytes, hipStream_t stream);

int isic_gemm_f32p_rows_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const int* a_rows,
                               const float* B, int ldb, const int* b_rows, float* C, int ldc, const float* bias, int act,
                               float beta, void* workspace, size_t workspace_bytes, hipStream_t stream);
// register-fed split-K kernel for A^T B with a small output and a long reduction (gemm_f32t.hip)
size_t isic_gemm_f32t_workspace_bytes(int transA, int transB, int M, int N, int K);
int isic_gemm_f32t_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          float* C, int ldc, const float* bias, int act, float beta, void* workspace,
                          size_t workspace_bytes, int force, hipStream_t stream);

// row-panel kernel for a long M and K <= 128 (gemm_f32r.hip)
int isic_gemm_f32r_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          float* C, int ldc, const float* bias, int act, float beta, const float* addend, int ldadd,
                          hipStream_t stream);

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDT = BK + 4;  // 80-byte rows: 16-B aligned, conflict-light

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K, lda, ldb, ldc;
  int transA, transB, act, vecA, vecB;
  float beta;
  int ksplit, klen;     // blockIdx.z handles k in [z*klen, (z+1)*klen): partial sums are atomically added to a pre-scaled C
  float* partial;       // ... or, with a workspace, stored as [ksplit][M][N] and added in split order by gemm_split_reduce_kernel
};

// Load 4 consecutive elements (along the contiguous dim) of a row-major matrix
// with R rows x Cc cols, zero-filled outside.
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int ld, int r, int c, int R, int Cc, int vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r >= R) return v;
  const float* q = p + (size_t)r * ld + c;
  if (vec && c + 3 < Cc) return *reinterpret_cast<const float4*>(q);
  if (c < Cc) v.x = q[0];
  if (c + 1 < Cc) v.y = q[1];
  if (c + 2 < Cc) v.z = q[2];
  if (c + 3 < Cc) v.w = q[3];
  return v;
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging coordinates
  // k-contiguous source: thread -> (row = tid>>2, kq = (tid&3)*4)
  // row-contiguous source: thread -> (k = tid>>4, rq = (tid&15)*4)
  const int rowK = tid >> 2, kqK = (tid & 3) * 4;
  const int kR = tid >> 4, rqR = (tid & 15) * 4;

  const int kbeg = blockIdx.z * a.klen;                       // klen is a multiple of BK
  const int kend = min(a.K, kbeg + a.klen);
  const int nk = (kend - kbeg + BK - 1) / BK;
  float4 ra, rb;
  auto gload = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    if (!a.transA) ra = load4(a.A, a.lda, m0 + rowK, k0 + kqK, a.M, kend, a.vecA);
    else ra = load4(a.A, a.lda, k0 + kR, m0 + rqR, kend, a.M, a.vecA);
    if (a.transB) rb = load4(a.B, a.ldb, n0 + rowK, k0 + kqK, a.N, kend, a.vecB);
    else rb = load4(a.B, a.ldb, k0 + kR, n0 + rqR, kend, a.N, a.vecB);
  };
  auto lstore = [&]() {
    if (!a.transA) *reinterpret_cast<float4*>(&As[rowK * LDT + kqK]) = ra;
    else {
      As[(rqR + 0) * LDT + kR] = ra.x; As[(rqR + 1) * LDT + kR] = ra.y;
      As[(rqR + 2) * LDT + kR] = ra.z; As[(rqR + 3) * LDT + kR] = ra.w;
    }
    if (a.transB) *reinterpret_cast<float4*>(&Bs[rowK * LDT + kqK]) = rb;
    else {
      Bs[(rqR + 0) * LDT + kR] = rb.x; Bs[(rqR + 1) * LDT + kR] = rb.y;
      Bs[(rqR + 2) * LDT + kR] = rb.z; Bs[(rqR + 3) * LDT + kR] = rb.w;
    }
  };

  gload(0);
  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // previous tile fully consumed
    lstore();
    __syncthreads();
    if (kt + 1 < nk) gload(kt + 1);
    f32x4 af[2], bf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      af[i] = *reinterpret_cast<const f32x4*>(&As[(wm * 32 + i * 16 + fr) * LDT + 4 * fg]);
      bf[i] = *reinterpret_cast<const f32x4*>(&Bs[(wn * 32 + i * 16 + fr) * LDT + 4 * fg]);
    }
    // k-step outermost: consecutive MFMAs write different accumulators (a dependent one issues ~6 slots late)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
  }

  // epilogue: C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 32 + j * 16 + fr;
      if (col >= a.N) continue;
      const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + i * 16 + fg * 4 + r;
        if (row >= a.M) continue;
        float v = acc[i][j][r] + bv;
        if (a.act == ISIC_ACT_RELU) v = fmaxf(v, 0.f);
        else if (a.act == ISIC_ACT_TANH) v = isic_tanhf(v);
        float* cp = a.C + (size_t)row * a.ldc + col;
        if (a.ksplit > 1) {
          if (a.partial) a.partial[((size_t)blockIdx.z * a.M + row) * a.N + col] = v;      // deterministic: own slot
          else atomicAdd(cp, v);                               // C already holds beta * C (gemm_scale_kernel)
          continue;
        }
        if (a.beta != 0.f) v += a.beta * (*cp);
        *cp = v;
      }
    }
}

// C = beta * C + sum over the splits in a FIXED order (split-K happens without bias / activation only): 16 lanes per
// element (lane g adds splits g, g + 16, ...) joined by a fixed xor tree -- up to ~200 splits of a 128 x 128 output would
// otherwise be 200 dependent loads per thread
__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(const float* __restrict__ partial, int splits,
                                                                 float* __restrict__ C, int M, int N, int ldc, float beta) {
  const int64_t n = (int64_t)M * N;
  const int g = threadIdx.x & 15;
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  float s = 0.f;
  if (i < n)
    for (int z = g; z < splits; z += 16) s += partial[(size_t)z * n + i];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
  if (i < n && g == 0) {
    float* p = C + (i / N) * ldc + (i % N);
    *p = beta != 0.f ? beta * (*p) + s : s;
  }
}

// ... four consecutive elements per thread (N % 4 == 0, 16-byte aligned rows): 16 float4 columns x 16 split groups per block, a
// thread adds splits g, g + 16, ... with every load independent (16 in flight at 256 splits), the groups are joined through
// LDS in group order.  The 16-lane form above moves a 256 x 64 KB stack of partial tiles in 9.4 us; this one in half.
__global__ __launch_bounds__(256) void gemm_split_reduce4_kernel(const float* __restrict__ partial, int splits,
                                                                  float* __restrict__ C, int M, int N, int ldc, float beta) {
  __shared__ f32x4 red[16][16];
  const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t n4 = (int64_t)M * N / 4, e4 = (int64_t)blockIdx.x * 16 + q;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (e4 < n4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(partial) + e4;
    for (int z = g; z < splits; z += 16) s += p[(size_t)z * n4];
  }
  red[g][q] = s;
  __syncthreads();
  if (threadIdx.x < 16 && e4 < n4) {
    f32x4 t = red[0][q];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][q];
    const int64_t i = e4 * 4;
    f32x4* out = reinterpret_cast<f32x4*>(C + (i / N) * ldc + (i % N));
    *out = beta != 0.f ? beta * (*out) + t : t;
  }
}

// C[M,N] *= beta (0: zero fill) ahead of a split-K accumulation
__global__ void gemm_scale_kernel(float* __restrict__ C, int M, int N, int ldc, float beta) {
  const int64_t n = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float* p = C + (i / N) * ldc + (i % N);
    *p = beta == 0.f ? 0.f : beta * (*p);
  }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int M, int N, int ldx,
                                                      float* __restrict__ out, float beta, int rows_per_block) {
  // one block per 64 columns x row chunk; 4 waves stride the rows; deterministic tree over waves.  With more than
  // one row chunk (gridDim.y > 1) the chunk sums are atomically added to the pre-scaled output.
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + wave; r < r1; r += 4) s += X[(size_t)r * ldx + col];
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < N) {
    float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (gridDim.y > 1) atomicAdd(out + col, t);
    else out[col] = beta != 0.f ? beta * out[col] + t : t;
  }
}

// A SHORT matrix (M < 8192 rows: bias gradients over the bags / graphs of a batch, the per-bag sums of the attention pool): 16
// waves stride the rows with four loads in flight each -- the 4-wave kernel above walks 64 dependent row loads per wave for
// 256 rows (12 us; this one 3) -- and are added in wave order: bit-reproducible.
__global__ __launch_bounds__(1024) void colsum_small_kernel(const float* __restrict__ X, int M, int N, int ldx,
                                                             float* __restrict__ out, float beta) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < N) {
    const float* p = X + col;
    int r = wave;
    for (; r + 48 < M; r += 64) {
      s0 += p[(size_t)r * ldx]; s1 += p[(size_t)(r + 16) * ldx]; s2 += p[(size_t)(r + 32) * ldx]; s3 += p[(size_t)(r + 48) * ldx];
    }
    for (; r < M; r += 16) s0 += p[(size_t)r * ldx];
  }
  part[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && col < N) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part[w][lane];
    out[col] = beta != 0.f ? beta * out[col] + t : t;
  }
}

// Column sums of a tall matrix (bias gradients over all the nodes of a batch: M ~ 50 k rows, N = 128..512): 16-byte
// loads (a thread owns 4 consecutive columns), VGB vector groups x 1024/VGB row lanes per block, four rows in flight per
// thread; LDS tree over the row lanes, one atomic per column and block into the pre-scaled output.  FEW, FAT blocks:
// same-address atomics retire at ~3 ns each per cache line (measured: 392 blocks x 128 columns took 41 us, all of it
// atomics), so the grid is ~96 blocks of 16 waves, not one block per 128 rows.
// With `partial` (workspace [gridDim.y][N]) a block stores its chunk sums there and colsum_reduce_kernel adds the chunks in
// order: bit-reproducible.  Without a workspace the chunks meet through fp32 atomics in arrival order.
template <int VGB>
__global__ __launch_bounds__(1024) void colsum4_kernel(const float* __restrict__ X, int M, int N, int ldx,
                                                       float* __restrict__ out, int rows_per_block,
                                                       float* __restrict__ partial) {
  constexpr int RL = 1024 / VGB;
  __shared__ f32x4 part[1024];
  const int vg = threadIdx.x % VGB, rl = threadIdx.x / VGB;
  const int col = (blockIdx.x * VGB + vg) * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (col < N) {
    const float* p = X + col;
    int r = r0 + rl;
    for (; r + 3 * RL < r1; r += 4 * RL) {
      const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)r * ldx));
      const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)(r + RL) * ldx));
      const f32x4 c = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)(r + 2 * RL) * ldx));
      const f32x4 d = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)(r + 3 * RL) * ldx));
      s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; r < r1; r += RL) s0 += *reinterpret_cast<const f32x4*>(p + (size_t)r * ldx);
  }
  part[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && col < N) {
    f32x4 t = part[vg];
#pragma unroll
    for (int k = 1; k < RL; ++k) t += part[k * VGB + vg];
    if (partial) {
      *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.y * N + col) = t;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(out + col + e, t[e]);
    }
  }
}

// out[n] = beta * out[n] + sum over the chunks in a fixed order: 16 lanes per column (lane g adds chunks g, g + 16, ...)
// joined by a fixed xor tree (one thread per column walked ~96 dependent loads: 19 us for 128 columns)
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ partial, int chunks, int N,
                                                             float* __restrict__ out, float beta) {
  const int g = threadIdx.x & 15;
  const int n = blockIdx.x * 16 + (threadIdx.x >> 4);
  float s = 0.f;
  if (n < N)
    for (int c = g; c < chunks; c += 16) s += partial[(size_t)c * N + n];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
  if (n < N && g == 0) out[n] = beta != 0.f ? beta * out[n] + s : s;
}

__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ t, float* __restrict__ dx,
                                int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float tv = t[i];
    dx[i] = dy[i] * (1.f - tv * tv);
  }
}

__global__ void relu_dropout_fwd_kernel(float* __restrict__ x, int64_t n, unsigned int thr, float scale,
                                        unsigned long long seed, unsigned long long stream_id,
                                        const unsigned long long* __restrict__ clock) {
  if (clock) stream_id += clock[0] * 1024ULL;              // device step clock (captured graphs)
  // each thread handles one Philox block = 4 consecutive elements
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nb = (n + 3) >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  for (; b < nb; b += stride) {
    Philox4 r = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (thr) r = philox_block((unsigned long long)b, seed, stream_id);
    const unsigned int w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = b * 4 + j;
      if (i < n) {
        float v = fmaxf(x[i], 0.f);
        if (thr) v = (w[j] >= thr) ? v * scale : 0.f;
        x[i] = v;
      }
    }
  }
}

__global__ void relu_dropout_bwd_kernel(const float* __restrict__ y, float* __restrict__ dy, int64_t n,
                                        float scale) {
  // y = dropout(relu(x)): y > 0 <=> x > 0 and kept, so the mask is recovered from y itself
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dy[i] = y[i] > 0.f ? dy[i] * scale : 0.f;
}

// C[M,N] += X[M,N] (row-major with leading dimensions): the unfused form of isic_gemm_f32_add_ws's addend
__global__ void add2d_kernel(float* __restrict__ C, int ldc, const float* __restrict__ X, int ldx, int M, int N) {
  const int64_t n = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / N, c = i - r * N;
    C[r * ldc + c] += X[r * ldx + c];
  }
}

// up to 32 (dst, src, count) segments in one launch: blockIdx.y = segment
struct MultiCopyArgs { float* dst[32]; const float* src[32]; long long count[32]; int accumulate; };
__global__ __launch_bounds__(256) void multi_copy_kernel(MultiCopyArgs a) {
  const int sgm = blockIdx.y;
  float* __restrict__ d = a.dst[sgm];
  const float* __restrict__ s = a.src[sgm];
  const long long n = a.count[sgm];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    d[i] = a.accumulate ? d[i] + s[i] : s[i];
}

__global__ void relu_dropout_bwd_out_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                            int64_t n, float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dx[i] = y[i] > 0.f ? dy[i] * scale : 0.f;
}

// split K when a long reduction meets a small output (weight gradients dW = dY^T X over all the nodes of a batch):
// a handful of 64x64 tiles would otherwise walk tens of thousands of k on a handful of CUs
inline int small_split_plan(int M, int N, int K, bool plain, int* klen) {
  *klen = ((K + BK - 1) / BK) * BK;
  const long long tiles = (long long)ceil_div(M, BM) * ceil_div(N, BN);
  // (a handful of tiles walking even 256 k is latency-bound -- 16 K-tiles of two barriers and an exposed global load each:
  //  22-25 us for the classifier's dW = dY^T X over a 256-graph batch; 64 k per split brings it to the launch floor)
  const bool tiny = plain && tiles <= 8 && K >= 256 && K < 2048;
  if (!plain || (K < 2048 && !tiny) || tiles >= 512) return 1;
  long long want = (1024 + tiles - 1) / tiles;                     // ~4 blocks per CU
  const long long max_split = tiny ? K / 64 : K / 256;             // at least 256 (64) k per split
  if (want > max_split) want = max_split;
  if (want <= 1) return 1;
  *klen = (int)(((K + want - 1) / want + BK - 1) / BK) * BK;
  return ceil_div(K, *klen);
}

inline int grid_for(int64_t n, int block) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

void isic_gemm_split_reduce_launch(const float* partial, int splits, float* C, int M, int N, int ldc, float beta,
                                   hipStream_t stream) {
  if (N % 4 == 0 && ldc % 4 == 0 && ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(C)) & 15) == 0) {
    hipLaunchKernelGGL(gemm_split_reduce4_kernel, dim3((unsigned)ceil_div64((int64_t)M * N / 4, 16)), dim3(256), 0, stream,
                       partial, splits, C, M, N, ldc, beta);
    return;
  }
  hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3((unsigned)ceil_div64((int64_t)M * N, 16)), dim3(256), 0, stream, partial,
                     splits, C, M, N, ldc, beta);
}

extern "C" {

size_t isic_gemm_f32_workspace_bytes(int transA, int transB, int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const size_t t = isic_gemm_f32t_workspace_bytes(transA, transB, M, N, K);
  const size_t p = isic_gemm_f32p_workspace_bytes(transA, transB, M, N, K);
  // the 64 x 64 kernel's own split-K (small_split_plan: a long reduction onto < 512 tiles, no bias / activation)
  int klen = 0;
  const int ks = small_split_plan(M, N, K, true, &klen);
  const size_t q = ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
  const size_t pq = p > q ? p : q;
  return t > pq ? t : pq;                                    // whichever kernel the launch ends up with
}

int isic_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                  float* C, int ldc, const float* bias, int act, float beta, void* stream) {
  return isic_gemm_f32_ws(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, beta, nullptr, 0, stream);
}

int isic_gemm_f32_ws(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int act, float beta, void* workspace, size_t workspace_bytes,
                     void* stream) {
  return isic_test_gemm_f32_variant(0, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, beta, workspace,
                                    workspace_bytes, stream);
}

int isic_gemm_f32_rows_ws(int transA, int transB, int M, int N, int K, const float* A, int lda, const int32_t* a_rows,
                          const float* B, int ldb, const int32_t* b_rows, float* C, int ldc, const float* bias, int act,
                          float beta, void* workspace, size_t workspace_bytes, void* stream) {
  if (!a_rows && !b_rows)
    return isic_gemm_f32_ws(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, beta, workspace, workspace_bytes, stream);
  ISIC_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C);
  ISIC_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N);
  ISIC_CHECK_ARG(act >= ISIC_ACT_NONE && act <= ISIC_ACT_TANH);
  return isic_gemm_f32p_rows_launch(transA, transB, M, N, K, A, lda, a_rows, B, ldb, b_rows, C, ldc, bias, act, beta, workspace,
                                    workspace_bytes, as_stream(stream));      // ISIC_ERR_UNSUPPORTED: gather first, then isic_gemm_f32_ws
}

int isic_gemm_f32_add_ws(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, const float* bias, int act, float beta, const float* addend, int ldadd,
                         void* workspace, size_t workspace_bytes, void* stream) {
  if (!addend)
    return isic_gemm_f32_ws(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, beta, workspace, workspace_bytes, stream);
  ISIC_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && ldadd >= N);
  if (M == 0 || N == 0) return ISIC_OK;
  ISIC_CHECK_ARG(A && B && C);
  ISIC_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N);
  ISIC_CHECK_ARG(act >= ISIC_ACT_NONE && act <= ISIC_ACT_TANH);
  int rc = isic_gemm_f32r_launch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, beta, addend, ldadd, as_stream(stream));
  if (rc != ISIC_ERR_UNSUPPORTED) return rc;             // the row-panel kernel adds it in its epilogue
  rc = isic_gemm_f32_ws(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, beta, workspace, workspace_bytes, stream);
  if (rc != ISIC_OK) return rc;
  hipLaunchKernelGGL(add2d_kernel, dim3(grid_for((int64_t)M * N, 256)), dim3(256), 0, as_stream(stream), C, ldc, addend, ldadd, M, N);
  return isic_launch_status();
}

int isic_test_gemm_f32_variant(int variant, int transA, int transB, int M, int N, int K, const float* A, int lda,
                               const float* B, int ldb, float* C, int ldc, const float* bias, int act, float beta,
                               void* workspace, size_t workspace_bytes, void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && variant >= 0 && variant <= 4);
  if (M == 0 || N == 0) return ISIC_OK;
  ISIC_CHECK_ARG(A && B && C);
  ISIC_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N);
  ISIC_CHECK_ARG(act >= ISIC_ACT_NONE && act <= ISIC_ACT_TANH);
  if (variant == 0 || variant == 3) {
    const int rc = isic_gemm_f32t_launch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, beta, workspace,
                                         workspace_bytes, variant == 3, as_stream(stream));
    if (rc != ISIC_ERR_UNSUPPORTED || variant == 3) return rc;
  }
  if (variant == 0 || variant == 4) {
    const int rc = isic_gemm_f32r_launch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, beta, nullptr, 0,
                                         as_stream(stream));
    if (rc != ISIC_ERR_UNSUPPORTED || variant == 4) return rc;
  }
  if (variant != 1 && variant != 4) {
    const int rc = isic_gemm_f32p_launch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, beta, workspace,
                                         workspace_bytes, as_stream(stream));
    if (rc != ISIC_ERR_UNSUPPORTED || variant == 2) return rc;   // launched (or failed for real): large, 16-byte friendly products
  }
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.transA = transA; a.transB = transB; a.act = act; a.beta = beta;
  a.vecA = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  a.vecB = ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  dim3 grid(ceil_div(M, BM), ceil_div(N, BN));
  ISIC_CHECK_ARG(grid.y <= 65535u);
  a.ksplit = small_split_plan(M, N, K, !bias && act == ISIC_ACT_NONE, &a.klen);
  a.partial = nullptr;
  if (a.ksplit > 1) {
    if (workspace && workspace_bytes >= (size_t)a.ksplit * M * N * sizeof(float))
      a.partial = reinterpret_cast<float*>(workspace);           // deterministic: per-split partials, added in order below
    else if (beta != 1.f)
      hipLaunchKernelGGL(gemm_scale_kernel, dim3(grid_for((int64_t)M * N, 256)), dim3(256), 0, as_stream(stream), C, M, N,
                         ldc, beta);
    grid.z = a.ksplit;
  }
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, as_stream(stream), a);
  if (a.partial) isic_gemm_split_reduce_launch(a.partial, a.ksplit, C, M, N, ldc, beta, as_stream(stream));
  return isic_launch_status();
}

int isic_multi_copy_f32(int nseg, float* const* dst, const float* const* src, const int64_t* count, int accumulate,
                        void* stream) {
  ISIC_CHECK_ARG(nseg >= 0 && nseg <= 32);
  if (nseg == 0) return ISIC_OK;
  ISIC_CHECK_ARG(dst && src && count);
  MultiCopyArgs a;
  long long mx = 0;
  for (int i = 0; i < 32; ++i) {
    a.dst[i] = i < nseg ? dst[i] : nullptr; a.src[i] = i < nseg ? src[i] : nullptr; a.count[i] = i < nseg ? count[i] : 0;
    if (i < nseg) { ISIC_CHECK_ARG(count[i] >= 0 && (count[i] == 0 || (dst[i] && src[i]))); if (count[i] > mx) mx = count[i]; }
  }
  a.accumulate = accumulate;
  if (mx == 0) return ISIC_OK;
  hipLaunchKernelGGL(multi_copy_kernel, dim3(grid_for(mx, 256) > 64 ? 64 : grid_for(mx, 256), nseg), dim3(256), 0, as_stream(stream), a);
  return isic_launch_status();
}

size_t isic_colsum_f32_workspace_bytes(int M, int N) {
  if (M < 1024 || N <= 0 || N % 4 != 0) return 0;
  return (size_t)96 * N * sizeof(float) + 256;             // at most 96 row chunks (colsum4_kernel's launch shape)
}

int isic_colsum_f32(const float* X, int M, int N, int ldx, float* out, float beta, void* stream) {
  return isic_colsum_f32_ws(X, M, N, ldx, out, beta, nullptr, 0, stream);
}

int isic_colsum_f32_ws(const float* X, int M, int N, int ldx, float* out, float beta, void* workspace,
                       size_t workspace_bytes, void* stream) {
  ISIC_CHECK_ARG(M >= 0 && N >= 0 && ldx >= N);
  if (N == 0) return ISIC_OK;
  ISIC_CHECK_ARG(X && out);
  if (M >= 1024 && N % 4 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
    const int nv = N / 4;
    const int vgb = nv >= 64 ? 64 : nv > 16 ? 32 : 16;
    const int colblocks4 = ceil_div(nv, vgb);
    int chunks4 = ceil_div(96, colblocks4);                        // few fat blocks: see colsum4_kernel
    const int min_rows = 4 * (1024 / vgb) * 2;                     // at least two unrolled iterations per thread
    if (chunks4 > M / min_rows) chunks4 = M / min_rows;
    if (chunks4 < 1) chunks4 = 1;
    const int rpb = ceil_div(M, chunks4);
    chunks4 = ceil_div(M, rpb);
    float* partial = nullptr;                                      // deterministic path: chunk sums parked in the workspace
    if (workspace && workspace_bytes >= (size_t)chunks4 * N * sizeof(float) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0)
      partial = reinterpret_cast<float*>(workspace);
    if (!partial)
      hipLaunchKernelGGL(gemm_scale_kernel, dim3(1), dim3(256), 0, as_stream(stream), out, 1, N, N, beta);
    const dim3 grid(colblocks4, chunks4);
    if (vgb == 64) hipLaunchKernelGGL(colsum4_kernel<64>, grid, dim3(1024), 0, as_stream(stream), X, M, N, ldx, out, rpb, partial);
    else if (vgb == 32) hipLaunchKernelGGL(colsum4_kernel<32>, grid, dim3(1024), 0, as_stream(stream), X, M, N, ldx, out, rpb, partial);
    else hipLaunchKernelGGL(colsum4_kernel<16>, grid, dim3(1024), 0, as_stream(stream), X, M, N, ldx, out, rpb, partial);
    if (partial)
      hipLaunchKernelGGL(colsum_reduce_kernel, dim3(ceil_div(N, 16)), dim3(256), 0, as_stream(stream), partial, chunks4, N, out,
                         beta);
    return isic_launch_status();
  }
  // few column blocks x many rows (bias gradients over all the nodes of a batch): split the rows over blockIdx.y
  int chunks = 1;
  const int colblocks = ceil_div(N, 64);
  if (M >= 8192 && colblocks < 256) {
    chunks = (1024 + colblocks - 1) / colblocks;
    if (chunks > M / 1024) chunks = M / 1024;
    if (chunks < 1) chunks = 1;
  }
  const int rows_per_block = ceil_div(M > 0 ? M : 1, chunks);
  chunks = ceil_div(M > 0 ? M : 1, rows_per_block);
  if (chunks == 1) {
    hipLaunchKernelGGL(colsum_small_kernel, dim3(colblocks), dim3(1024), 0, as_stream(stream), X, M, N, ldx, out, beta);
    return isic_launch_status();
  }
  if (chunks > 1)
    hipLaunchKernelGGL(gemm_scale_kernel, dim3(1), dim3(256), 0, as_stream(stream), out, 1, N, N, beta);
  hipLaunchKernelGGL(colsum_kernel, dim3(colblocks, chunks), dim3(256), 0, as_stream(stream), X, M, N, ldx, out, beta,
                     rows_per_block);
  return isic_launch_status();
}

int isic_tanh_bwd_f32(const float* dy, const float* t, float* dx, int64_t n, void* stream) {
  ISIC_CHECK_ARG(n >= 0);
  if (n == 0) return ISIC_OK;
  ISIC_CHECK_ARG(dy && t && dx);
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), dy, t, dx, n);
  return isic_launch_status();
}

int isic_relu_dropout_fwd_f32(float* x, int64_t n, uint32_t drop_threshold, float drop_scale, uint64_t seed,
                              uint64_t stream_id, void* stream) {
  return isic_relu_dropout_fwd_clk_f32(x, n, drop_threshold, drop_scale, seed, stream_id, nullptr, stream);
}

int isic_relu_dropout_fwd_clk_f32(float* x, int64_t n, uint32_t drop_threshold, float drop_scale, uint64_t seed,
                                  uint64_t stream_id, const uint64_t* clock, void* stream) {
  ISIC_CHECK_ARG(n >= 0);
  if (n == 0) return ISIC_OK;
  ISIC_CHECK_ARG(x);
  hipLaunchKernelGGL(relu_dropout_fwd_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), x, n,
                     drop_threshold, drop_scale, (unsigned long long)seed, (unsigned long long)stream_id,
                     (const unsigned long long*)clock);
  return isic_launch_status();
}

int isic_relu_dropout_bwd_out_f32(const float* y, const float* dy, float* dx, int64_t n, float drop_scale, void* stream) {
  ISIC_CHECK_ARG(n >= 0);
  if (n == 0) return ISIC_OK;
  ISIC_CHECK_ARG(y && dy && dx);
  hipLaunchKernelGGL(relu_dropout_bwd_out_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), y, dy, dx, n, drop_scale);
  return isic_launch_status();
}

int isic_relu_dropout_bwd_f32(const float* y, float* dy, int64_t n, float drop_scale, void* stream) {
  ISIC_CHECK_ARG(n >= 0);
  if (n == 0) return ISIC_OK;
  ISIC_CHECK_ARG(y && dy);
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), y, dy, n,
                     drop_scale);
  return isic_launch_status();
}

}  // extern "C"
