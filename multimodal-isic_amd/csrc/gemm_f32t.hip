// Weight-gradient GEMM of the graph / MIL heads:  C[M,N] = beta * C + A^T B  with A = dY [K x M], B = X [K x N], both row-major,
// a SMALL output (M, N multiples of 64: 128 x 128 .. 128 x 768) and a LONG reduction over all the nodes of a batch
// (K = 50,176 at BASELINE.json configs[3]).  Exact fp32 on v_mfma_f32_16x16x4_f32 (gfx950).
//
// The product is bound twice at once -- 2 K (M + N) * 2 bytes of HBM against 2 M N K flops: 51 MB / 1.64 GFLOP for the
// 128 x 128 layers, 10 us either way -- so nothing may be staged twice.  No LDS for the operands at all:
//   * both operands have the OUTPUT index contiguous, which is what the MFMA wants per lane if the rows of a 16 x 16 tile
//     are allowed to be any 16 rows: lane l loads ONE float4 A[k0 + (l >> 4)][m0 + 4 (l & 15) ..+3]; component r of it
//     is the A operand (row i = l & 15, k = l >> 4) of the tile whose rows are m0 + 4 i + r.  One 1 KB load of A and one
//     of B per k-step of 4 feed 4 x 4 MFMAs (a 64 x 64 wave tile) -- 16 B per lane, fully coalesced, every byte used once
//     per wave;
//   * a block is a 128 x 128 output tile x a slab of K: 2 x 2 wave tiles x 2 K-groups (the two groups walk alternate
//     k-steps, their sums are joined through LDS at the end); P k-steps of loads are in flight per wave (registers);
//   * all 256 CUs get a slab; the slabs' partial tiles go to the caller's workspace and are added in slab order by
//     gemm_split_reduce_kernel (gemm_f32.hip): no atomics, bit-reproducible.
// Replaces, for these shapes, the 64 x 64 x 16 kernel's split-K (196 splits of 256 k, k-strided operands transposed through
// scalar LDS stores): 85 us -> see profiles/r03_gemm_bench.txt.  Reference: the autograd backward of the nn.Linear /
// GCNConv weights in 05_train_gnns.py:66,82,126-139.
#include "common.h"

namespace {

constexpr int TP = 8;                    // k-steps of loads in flight per wave

struct TnArgs {
  const float* A;
  const float* B;
  float* partial;                        // [slabs][M][N]
  int M, N, K, lda, ldb;
  int tiles_n, tiles, slabs, steps_per_slab, total_steps;     // a k-step = 4 rows
};

__global__ __launch_bounds__(512) void gemm_tn_skinny_kernel(TnArgs a) {
  __shared__ f32x4 join[4][16][64];      // K-group 1's accumulators: [wave tile][acc][lane]  (64 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, wq = wave & 3, wm = wq >> 1, wn = wq & 1;
  const int tile = blockIdx.x / a.slabs, slab = blockIdx.x - tile * a.slabs;      // the tiles of a slab sit `slabs` apart: same XCD
  const int tm = tile / a.tiles_n, tn = tile - tm * a.tiles_n;
  const int m0 = tm * 128 + wm * 64, n0 = tn * 128 + wn * 64;
  const bool live = m0 < a.M && n0 < a.N;                    // wave-uniform (M, N are multiples of 64)
  const int s_begin = slab * a.steps_per_slab, s_end = min(a.total_steps, s_begin + a.steps_per_slab);

  f32x4 acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (live) {
    // The operand ring is requested with inline-asm loads hipcc does not track and counted by hand (as gemm_f32r.hip; its own
    // s_waitcnt insertion drained the ring -- vmcnt(0) -- once per TP k-steps, right after a refill: half the launch).  A slot
    // is requested 2 loads at a time, in order, so "slot p has landed" = at most 2 (TP - 1) younger requests in flight.
    // Rows past K are clamped (finite values, multiplied by nothing: the loop stops at s_end; the ring's overrun is not used).
    const int kr = lane >> 4, c4 = (lane & 15) * 4;
    const float* pa = a.A + (size_t)m0 + c4;
    const float* pb = a.B + (size_t)n0 + c4;
    f32x4 fa[TP], fb[TP];
    auto request = [&](int slot, int s) {                    // k-step s of this wave's K-group: rows 4 s .. 4 s + 3
      const int k = min(4 * s + kr, a.K - 1);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fa[slot]) : "v"(pa + (size_t)k * a.lda) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fb[slot]) : "v"(pb + (size_t)k * a.ldb) : "memory");
    };
    const bool ktail = (a.K & 3) != 0;                       // the last k-step has rows past K: their products must not count
    int s = s_begin + kg;                                    // the two K-groups walk alternate k-steps
#pragma unroll
    for (int p = 0; p < TP; ++p) request(p, s + 2 * p);
    for (; s < s_end; s += 2 * TP) {
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        if (s + 2 * p >= s_end) break;                       // wave-uniform: no multiplies past the slab
        // wait, THEN copy the slot out -- inside one asm statement: a tied in/out operand made the compiler copy the slot
        // in front of the wait (operand set-up), i.e. before it had landed
        f32x4 va, vb;
        asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b64 %0, %2\n\tv_mov_b64 %1, %3"
                     : "=&v"(*reinterpret_cast<unsigned long long*>(&va)), "=&v"(*(reinterpret_cast<unsigned long long*>(&va) + 1))
                     : "v"(*reinterpret_cast<const unsigned long long*>(&fa[p])),
                       "v"(*(reinterpret_cast<const unsigned long long*>(&fa[p]) + 1)), "n"(2 * (TP - 1)) : "memory");
        asm volatile("v_mov_b64 %0, %2\n\tv_mov_b64 %1, %3"
                     : "=&v"(*reinterpret_cast<unsigned long long*>(&vb)), "=&v"(*(reinterpret_cast<unsigned long long*>(&vb) + 1))
                     : "v"(*reinterpret_cast<const unsigned long long*>(&fb[p])),
                       "v"(*(reinterpret_cast<const unsigned long long*>(&fb[p]) + 1)) : "memory");
        if (ktail && 4 * (s + 2 * p) + kr >= a.K) va = (f32x4){0.f, 0.f, 0.f, 0.f};       // (a clamped row is a repeated row)
        request(p, s + 2 * (p + TP));                        // refill the slot: TP k-steps ahead
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[r], vb[c], acc[r][c], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the ring's overrun lands before the registers are reused
  }

  // join the two K-groups (fixed order: group 0 + group 1), then this slab's partial tile
  if (kg == 1 && live) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) join[wq][r * 4 + c][lane] = acc[r][c];
  }
  __syncthreads();
  if (kg == 0 && live) {
    // acc[r][c][v] of lane l = C[m0 + 16 (l >> 4) + 4 v + r][n0 + 4 (l & 15) + c]
    float* out = a.partial + ((size_t)slab * a.M + m0 + 16 * (lane >> 4)) * a.N + n0 + 4 * (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 t[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) t[c] = acc[r][c] + join[wq][r * 4 + c][lane];
#pragma unroll
      for (int v = 0; v < 4; ++v)
        *reinterpret_cast<f32x4*>(out + (size_t)(4 * v + r) * a.N) = (f32x4){t[0][v], t[1][v], t[2][v], t[3][v]};
    }
  }
}

struct TnPlan { int tiles_m, tiles_n, tiles, slabs, steps_per_slab, total_steps; };

bool tn_plan(int transA, int transB, int M, int N, int K, int lda, int ldb, const void* A, const void* B, bool force, TnPlan& p) {
  if (!transA || transB || M % 64 != 0 || N % 64 != 0 || K < 4096) return false;
  if ((lda & 3) || (ldb & 3) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15)) return false;
  if (!force && (int64_t)M * N > 128 * 256) return false;   // measured (profiles/r03_gemm_bench.txt): it wins on one or two 128 x 128
  if ((int64_t)M * N > 128 * 1024) return false;            // tiles; from 512 x 128 on the persistent LDS-DMA kernel is as fast
  const int cus = isic_cu_count();
  p.tiles_m = ceil_div(M, 128); p.tiles_n = ceil_div(N, 128); p.tiles = p.tiles_m * p.tiles_n;
  p.total_steps = ceil_div(K, 4);
  int slabs = cus / p.tiles;
  if (slabs >= 8) slabs &= ~7;                              // the tiles of a slab on one XCD (block index = tile * slabs + slab)
  if (slabs < 1) slabs = 1;
  p.steps_per_slab = ceil_div(p.total_steps, slabs);
  if (p.steps_per_slab < 2 * TP) p.steps_per_slab = 2 * TP; // at least one full ring per K-group
  p.slabs = ceil_div(p.total_steps, p.steps_per_slab);
  return p.slabs >= 2;
}

}  // namespace

// gemm_f32.hip
void isic_gemm_split_reduce_launch(const float* partial, int splits, float* C, int M, int N, int ldc, float beta,
                                   hipStream_t stream);

size_t isic_gemm_f32t_workspace_bytes(int transA, int transB, int M, int N, int K) {
  TnPlan p;
  if (!tn_plan(transA, transB, M, N, K, 4, 4, nullptr, nullptr, true, p)) return 0;
  return (size_t)p.slabs * M * N * sizeof(float);
}

// ISIC_ERR_UNSUPPORTED: not a shape for this kernel (the caller falls through to the other kernels)
int isic_gemm_f32t_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          float* C, int ldc, const float* bias, int act, float beta, void* workspace,
                          size_t workspace_bytes, int force, hipStream_t stream) {
  TnPlan p;
  if (bias || act != ISIC_ACT_NONE || !tn_plan(transA, transB, M, N, K, lda, ldb, A, B, force != 0, p)) return ISIC_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < (size_t)p.slabs * M * N * sizeof(float)) return ISIC_ERR_UNSUPPORTED;
  TnArgs a;
  a.A = A; a.B = B; a.partial = reinterpret_cast<float*>(workspace);
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb;
  a.tiles_n = p.tiles_n; a.tiles = p.tiles; a.slabs = p.slabs; a.steps_per_slab = p.steps_per_slab; a.total_steps = p.total_steps;
  hipLaunchKernelGGL(gemm_tn_skinny_kernel, dim3(p.tiles * p.slabs), dim3(512), 0, stream, a);
  isic_gemm_split_reduce_launch(a.partial, p.slabs, C, M, N, ldc, beta, stream);
  return isic_launch_status();
}
