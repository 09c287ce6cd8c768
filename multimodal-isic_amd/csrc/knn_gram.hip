// k-NN adjacency build of the patch graphs (03_build_graphs.py:37-54 == utils_g_mil.py:596-615), the fast path of
// isic_knn_graph for graphs of up to 208 nodes (ResNet / ViT patch grids: 196):
//
//   d[i][j] = (|x_i|^2 + |x_j|^2) - 2 x_i . x_j      (03:46-47, fp32)      clamp(min = 0) (:48)      diag = +inf (:49)
//   nn[i][0..k) = the k smallest of row i, ascending, ties -> lower index (:50 topk(largest = False))
//
// The whole Gram matrix G = X X^T of a graph (196 x 196 x 768: 59 MFLOP) is formed by ONE workgroup on the exact-fp32 matrix
// core (v_mfma_f32_16x16x4_f32), and the top-16 of every row is selected out of the accumulator registers -- a distance never
// touches memory.  (The first kernel, knn_kernel of graph.hip, gave a block 16 query rows: every block re-read the graph's
// 602 KB of features out of L2 with per-lane global loads and ran at 0.17 of the fp32 matrix peak.)
//
//   * 13 waves (832 threads); wave r owns the row stripe 16r .. 16r+15 against ALL 13 column tiles: 13 tiles x 4 accumulator
//     VGPRs.  A lane therefore ends with rows 4fg + j (j = 0..3) x columns 16c + fr (c = 0..12): a row's 208 distances live
//     in the 16 lanes of one DPP row -- the selection is a local scan + a 4-step row rotation, no LDS, no shuffles.
//   * X is staged through LDS in K-chunks of 32 floats ([208 rows][128 B], 16-byte chunks XOR-swizzled on the source side so
//     that the ds_read_b128 fragment reads are conflict-free), three stages, by LDS-DMA issued by the multiplying waves
//     themselves (two 1 KB pieces per wave and chunk), one s_barrier per chunk (104 MFMAs per wave), hand-counted vmcnt.
//     The block is PERSISTENT over graphs and the chunk ring runs across graph boundaries: the next graph's first chunks
//     arrive while this graph's rows are being selected.
//   * |x_i|^2 is the DIAGONAL of G -- the same MFMA summation order as the dot products, so d[i][i] is exactly 0 before it
//     is set to +inf and d is symmetric bit for bit; the row norms need no pass of their own.
//   * summation order of x_i . x_j: identical to knn_kernel's (k-steps of 16, the four MFMAs of a step take k = 4 fg + j).
#include "common.h"

namespace {

constexpr int KG_ROWS = 208;                 // 13 tiles of 16 nodes
constexpr int KG_TILES = 13;
constexpr int KG_THREADS = KG_TILES * 64;
constexpr int KG_STAGE = KG_ROWS * 128;      // one K-chunk of 32 floats for every row
constexpr int KG_NST = 3;
constexpr int KG_LDS = KG_NST * KG_STAGE + KG_ROWS * 4;

struct KnnGramArgs {
  const float* x;
  const int64_t* offsets;
  int64_t* nn_idx;
  float* nn_dist;
  int G, D, k;
};

__device__ __forceinline__ void kg_glds16(const void* gsrc, unsigned lds_dst) {     // outside hipcc's vmcnt bookkeeping
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int CTRL>
__device__ __forceinline__ float kg_dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int kg_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

template <bool WITH_DIST>
__global__ __launch_bounds__(KG_THREADS) void knn_gram_kernel(KnnGramArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  float* sqn = reinterpret_cast<float*>(smem + KG_NST * KG_STAGE);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int nchunks = a.D >> 5;
  const int ngr = (a.G - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // graphs of this block: b, b + grid, ...
  const int total = ngr * nchunks;
  if (total <= 0) return;

  // ---- staging: piece t (= 0, 1) of this wave brings rows 16 wave + 8 t + r8; the lane fetches global 16-byte chunk
  //      p ^ r8 of its row (row & 7 == r8), so that LDS position p of a row holds chunk p ^ (row & 7)
  const int r8 = lane >> 3, p8 = lane & 7;
  const unsigned src_chunk = (unsigned)((p8 ^ r8) << 4);
  int ig = 0, ikc = 0;                                      // (graph ordinal, K-chunk) of the next chunk to ISSUE
  auto issue = [&](int t) {
    const int g = (int)blockIdx.x + ig * (int)gridDim.x;
    const long long lo = a.offsets[g];
    const int N = (int)(a.offsets[g + 1] - lo);
    const unsigned dst = lds0 + (unsigned)((t % KG_NST) * KG_STAGE + wave * 2048);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int row = wave * 16 + h * 8 + r8;
      row = row < N ? row : N - 1;                         // rows past the graph repeat its last node (masked where they are used)
      row = row < 0 ? 0 : row;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(a.x + (size_t)(N > 0 ? lo + row : 0) * a.D) +
                                 (size_t)ikc * 128 + src_chunk;
      kg_glds16(src, dst + h * 1024);
    }
    if (++ikc == nchunks) { ikc = 0; ++ig; }
  };
  issue(0);
  if (total > 1) issue(1);

  // ---- fragment addresses inside a stage: row 16 c + fr, k-quarter kq (16 floats): chunk (4 kq + fg) ^ (fr & 7)
  const unsigned foff0 = (unsigned)(fr * 128 + ((fg ^ (fr & 7)) << 4));              // kq = 0; kq = 1: ^ 64
  f32x4 acc[KG_TILES];
#pragma unroll
  for (int c = 0; c < KG_TILES; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int cg = 0, ckc = 0;                                      // (graph ordinal, K-chunk) being multiplied
  for (int t = 0; t < total; ++t) {
    // chunk t has landed (this wave's pieces) when only the younger chunk's two pieces are still in flight.  The first chunk
    // of a graph follows the previous graph's result stores (issued BEHIND the pieces of chunks t and t + 1, which went out
    // before the selection and have long landed): everything is waited for there -- one store round trip per graph.
    if (t + 1 >= total || (ckc == 0 && t > 0)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // ... every wave's pieces; and stage (t + 2) % 3 is no longer read
    if (t + 2 < total) issue(t + 2);
    const unsigned char* st = smem + (t % KG_NST) * KG_STAGE;
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
      const unsigned char* q = st + (foff0 ^ (unsigned)(kq << 6));
      const f32x4 av = *reinterpret_cast<const f32x4*>(q + wave * 2048);
#pragma unroll
      for (int c0 = 0; c0 < KG_TILES; c0 += 4) {            // four column tiles at a time: a tile's accumulator is touched
        f32x4 bv[4];                                        // every fourth MFMA (a dependent MFMA issues ~6 slots late)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (c0 + i < KG_TILES) bv[i] = *reinterpret_cast<const f32x4*>(q + (c0 + i) * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (c0 + i < KG_TILES) acc[c0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[i][j], acc[c0 + i], 0, 0, 0);
      }
    }
    if (++ckc < nchunks) continue;

    // ================================================================ the graph is complete: distances -> top-k
    ckc = 0;
    const int g = (int)blockIdx.x + cg * (int)gridDim.x;
    ++cg;
    const long long lo = a.offsets[g];
    const int N = (int)(a.offsets[g + 1] - lo);
    // |x_i|^2 = G[i][i]: tile (wave, wave), row 4 fg + j == column fr
    {
      f32x4 dg = acc[0];
#pragma unroll
      for (int c = 1; c < KG_TILES; ++c) dg = wave == c ? acc[c] : dg;
      const float dv = (fr & 3) == 0 ? dg[0] : (fr & 3) == 1 ? dg[1] : (fr & 3) == 2 ? dg[2] : dg[3];
      if ((fr >> 2) == fg) sqn[wave * 16 + fr] = dv;
    }
    lds_barrier();
    if (wave * 16 < N) {
      float sr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) sr[j] = sqn[wave * 16 + fg * 4 + j];
#pragma unroll
      for (int c = 0; c < KG_TILES; ++c) {
        const int col = c * 16 + fr;
        const float sc = sqn[col];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = wave * 16 + fg * 4 + j;
          float dv = (sr[j] + sc) - 2.0f * acc[c][j];        // 03_build_graphs.py:47
          dv = fmaxf(dv, 0.f);                               // :48 clamp(min = 0)
          acc[c][j] = (col < N && col != q) ? dv : INFINITY; // :49 diagonal; columns past the graph
        }
      }
      int res_i[4];
      float res_v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { res_i[j] = -1; res_v[j] = INFINITY; }
      for (int round = 0; round < a.k; ++round) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float bv = acc[0][j];
          int bc = 0;
#pragma unroll
          for (int c = 1; c < KG_TILES; ++c) {               // strict <: the lowest column of this lane wins a tie
            const bool lt = acc[c][j] < bv;
            bv = lt ? acc[c][j] : bv;
            bc = lt ? c : bc;
          }
          int bi = bc * 16 + fr;
          // minimum of (value, index) over the 16 lanes of the DPP row: rotations by 8, 4, 2, 1
#define KG_STEP(CTRL)                                                        \
          {                                                                  \
            const float ov = kg_dpp_f<CTRL>(bv);                             \
            const int oi = kg_dpp_i<CTRL>(bi);                               \
            const bool take = ov < bv || (ov == bv && oi < bi);              \
            bv = take ? ov : bv;                                             \
            bi = take ? oi : bi;                                             \
          }
          KG_STEP(0x128) KG_STEP(0x124) KG_STEP(0x122) KG_STEP(0x121)
#undef KG_STEP
          const bool found = bv < INFINITY;
          if (fr == round) { res_i[j] = found ? bi : -1; res_v[j] = bv; }
          // the owner of the winner retires it
          const bool own = (bi & 15) == fr;
          const int cs = bi >> 4;
#pragma unroll
          for (int c = 0; c < KG_TILES; ++c) acc[c][j] = (own && cs == c) ? INFINITY : acc[c][j];
        }
      }
      // lane fr holds neighbour #fr of its four rows: one contiguous run of k indices per row
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = wave * 16 + fg * 4 + j;
        const bool ok = q < N && fr < a.k;
        int64_t* ip = a.nn_idx + (size_t)(ok ? (lo + q) * a.k + fr : 0);
        if (ok) *ip = (int64_t)res_i[j];
        if (WITH_DIST) {
          float* dp = a.nn_dist + (size_t)(ok ? (lo + q) * a.k + fr : 0);
          if (ok) *dp = res_v[j];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < KG_TILES; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}

}  // namespace

// fast path of isic_knn_graph (graph.hip): every graph has <= 208 nodes, D % 32 == 0, k <= 16
bool isic_knn_gram_supported(int D, int k, int max_nodes) {
  return D % 32 == 0 && D >= 32 && k >= 1 && k <= 16 && max_nodes >= 1 && max_nodes <= KG_ROWS;
}

int isic_knn_gram_launch(const float* x, const int64_t* offsets, int G, int D, int k, int64_t* nn_idx, float* nn_dist,
                         hipStream_t stream) {
  KnnGramArgs a;
  a.x = x; a.offsets = offsets; a.nn_idx = nn_idx; a.nn_dist = nn_dist; a.G = G; a.D = D; a.k = k;
  static IsicPerDeviceOnce once;
  if (isic_once_per_device(once, [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_gram_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, KG_LDS);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(knn_gram_kernel<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, KG_LDS);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  const int cus = isic_cu_count();
  // a whole number of graphs per block where possible (2048 graphs on 256 CUs: 8 each)
  const int per = ceil_div(G, cus);
  const int grid = ceil_div(G, per);
  if (nn_dist) hipLaunchKernelGGL(knn_gram_kernel<true>, dim3(grid), dim3(KG_THREADS), KG_LDS, stream, a);
  else hipLaunchKernelGGL(knn_gram_kernel<false>, dim3(grid), dim3(KG_THREADS), KG_LDS, stream, a);
  return ISIC_OK;
}
