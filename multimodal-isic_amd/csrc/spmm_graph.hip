// Graph-resident SpMM for a BATCH of small graphs (gfx950): the GCNConv aggregation of the patch-graph path
//   out[i,:] = alpha * sum_{e in row i} val[e] * x[col[e],:] (+ bias) (+ addend_scale * addend[i,:])
// (05_train_gnns.py:184-185, PyG GCNConv.propagate) when the operator is block-diagonal: graph g owns the node range
// [offsets[g], offsets[g+1]) and its edges stay inside it -- 256 graphs of 196 nodes per step in the reference geometry.
//
// The grouped-lane gather of graph.hip (spmm_group_kernel) reads every neighbour row out of L2: 9 x 512 B per output row
// at F = 128, 4.2x the compulsory bytes, and a wave walks rowptr -> col -> rows -> store as dependent round trips; it
// delivers the COMPULSORY bytes (x once, out once, the CSR) at 2.0-2.25 TB/s = 0.25-0.28 of the HBM peak, with the PMC
// counters showing that HBM itself moves no more than those compulsory bytes (profiles/r03_pmc_traffic.json).
// Here a graph's node features (196 x 128 x 4 B = 100 KB) are brought into LDS ONCE by LDS-DMA -- one 1024-thread block
// per graph, all sixteen waves issue the DMA -- while the same waves prefetch the CSR rows they will process; after one
// barrier the neighbour gather runs out of LDS (ds_read_b128, 16 consecutive lanes read 256 contiguous bytes:
// conflict-free) and only the output row goes back to memory.  HBM traffic = compulsory bytes, L2 traffic too.
// A neighbour outside the block's node range (never the case for a batch of graphs; kept for generality) is read from
// global memory instead.  Accumulation order per row is the edge order, as in spmm_group_kernel: bit-identical results.
//
// Round 2 tried a graph-resident variant and found it 2x SLOWER than the gather ("measured and rejected"): it
// staged 64-feature chunks through registers in a loop and ran one fat wave set per graph behind its own latencies.  The
// differences here: the whole graph in one DMA burst issued by all waves, the CSR prefetched under it, no chunk loop.
#include "common.h"

namespace {

constexpr int SG_WAVES = 16, SG_THREADS = SG_WAVES * 64;
constexpr int SG_MAX_ITERS = 8;                 // row groups a wave may own (prefetched CSR registers)
constexpr int SG_MAX_LDS = 152 * 1024;          // node features of one graph

struct SGArgs {
  const int* rowptr; const int* col; const float* val; const float* x; const float* bias; float* out;
  const int64_t* offsets;
  const float* addend;
  int F, max_nodes;             // max_nodes: the node count the launch's LDS was sized for
  float alpha, addend_scale;
};

__device__ __attribute__((aligned(256))) unsigned char g_sg_zero_page[1024];

__device__ __forceinline__ void sg_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// LPR lanes (16 bytes of the row each) own an output row, RW = 64 / LPR rows per wave pass; F <= 4 LPR.
template <int LPR>
__global__ __launch_bounds__(SG_THREADS) void spmm_graph_kernel(SGArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  constexpr int RW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gl = lane & (LPR - 1), g0 = lane - gl, sub = lane / LPR;
  const int F = a.F;
  const int64_t n0 = a.offsets[blockIdx.x], n1 = a.offsets[blockIdx.x + 1];
  const int rows = (int)(n1 - n0);
  if (rows <= 0) return;                                   // whole block
  // a graph larger than the launch was sized for (the caller's max_nodes was wrong) is still computed correctly: nothing
  // is staged and every neighbour comes from global memory
  const bool resident = rows <= a.max_nodes;
  // ---- phase 1a: the graph's node features -> LDS, row-major and dense: byte i of the graph at LDS byte i
  const long long total = (long long)rows * F * 4;         // <= SG_MAX_LDS when resident (host)
  if (resident) {
    const int chunks = (int)((total + 1023) / 1024);
    const unsigned char* gsrc = reinterpret_cast<const unsigned char*>(a.x + n0 * F);
    for (int c = wave; c < chunks; c += SG_WAVES) {
      const long long off = (long long)c * 1024 + lane * 16;
      // F % 4 == 0: a 16-byte piece is inside or outside the graph as a whole; outside -> zeros (never read back)
      const void* src = off < total ? (const void*)(gsrc + off) : (const void*)(g_sg_zero_page + lane * 16);
      sg_glds16(src, lds0 + (unsigned)c * 1024);
    }
  }
  // ---- phase 1b (under the DMA): this wave's CSR rows.  Row group it: rows (it * SG_WAVES + wave) * RW + sub
  const int groups = (rows + RW - 1) / RW;
  const int niter = (groups + SG_WAVES - 1) / SG_WAVES;
  int rb[SG_MAX_ITERS], rdeg[SG_MAX_ITERS], rc[SG_MAX_ITERS];
  float rwt[SG_MAX_ITERS];
#pragma unroll
  for (int it = 0; it < SG_MAX_ITERS; ++it) {
    const int r = (it * SG_WAVES + wave) * RW + sub;
    const bool ok = r < rows;
    rb[it] = ok ? a.rowptr[n0 + r] : 0;
    rdeg[it] = ok ? a.rowptr[n0 + r + 1] - rb[it] : 0;
  }
#pragma unroll
  for (int it = 0; it < SG_MAX_ITERS; ++it) {
    rc[it] = 0; rwt[it] = 0.f;
    if (gl < rdeg[it]) { rc[it] = a.col[rb[it] + gl]; rwt[it] = a.val[rb[it] + gl]; }      // the first LPR edges of the row
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's DMA pieces (and its CSR loads) have landed
  __syncthreads();                                         // ... every wave's

  // ---- phase 2: gather out of LDS
  const bool fl = gl * 4 < F;
  auto row_pass = [&](int r, int b, int deg, int c, float w) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0;; base += LPR) {
      const int cnt = min(LPR, deg - base);
      int maxcnt = cnt;                                    // wave-uniform trip count
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, o, 64));
      if (maxcnt <= 0) break;
      if (base > 0) {                                      // rows with more than LPR edges: the next LPR on demand
        c = 0; w = 0.f;
        if (gl < cnt) { c = a.col[b + base + gl]; w = a.val[b + base + gl]; }
      }
      for (int j = 0; j < maxcnt; ++j) {
        const int src = g0 + (j & (LPR - 1));
        const int cj = __shfl(c, src, 64);
        const float wj = __shfl(w, src, 64);
        if (j < cnt && fl) {
          const long long loc = (long long)cj - n0;
          float4 xv;
          if (resident && loc >= 0 && loc < rows) xv = *reinterpret_cast<const float4*>(smem + ((size_t)loc * F + gl * 4) * 4);
          else xv = *reinterpret_cast<const float4*>(a.x + (int64_t)cj * F + gl * 4);      // outside the graph: global
          acc.x += wj * xv.x; acc.y += wj * xv.y; acc.z += wj * xv.z; acc.w += wj * xv.w;
        }
      }
    }
    if (r < rows && fl) {
      const int64_t row = n0 + r;
      float4 o = make_float4(a.alpha * acc.x, a.alpha * acc.y, a.alpha * acc.z, a.alpha * acc.w);
      if (a.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(a.bias + gl * 4);
        o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
      }
      if (a.addend) {
        const float4 av = *reinterpret_cast<const float4*>(a.addend + row * F + gl * 4);
        o.x += a.addend_scale * av.x; o.y += a.addend_scale * av.y; o.z += a.addend_scale * av.z; o.w += a.addend_scale * av.w;
      }
      *reinterpret_cast<float4*>(a.out + row * F + gl * 4) = o;
    }
  };
#pragma unroll
  for (int it = 0; it < SG_MAX_ITERS; ++it) {
    if (it >= niter) break;
    row_pass((it * SG_WAVES + wave) * RW + sub, rb[it], rdeg[it], rc[it], rwt[it]);
  }
  for (int it = SG_MAX_ITERS; it < niter; ++it) {          // beyond the prefetched groups (oversized graph): on demand
    const int r = (it * SG_WAVES + wave) * RW + sub;
    const bool ok = r < rows;
    const int b = ok ? a.rowptr[n0 + r] : 0, deg = ok ? a.rowptr[n0 + r + 1] - b : 0;
    int c = 0;
    float w = 0.f;
    if (gl < deg) { c = a.col[b + gl]; w = a.val[b + gl]; }
    row_pass(r, b, deg, c, w);
  }
}

template <int LPR>
int launch_sg(const SGArgs& a, int n_graphs, int lds, hipStream_t s) {
  static IsicPerDeviceOnce once;              // hipFuncSetAttribute is per device (one flag set per template instance)
  if (isic_once_per_device(once, [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_graph_kernel<LPR>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SG_MAX_LDS + 1024);
      }) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL((spmm_graph_kernel<LPR>), dim3(n_graphs), dim3(SG_THREADS), lds, s, a);
  return ISIC_OK;
}

}  // namespace

extern "C" {

int isic_spmm_csr_graphs_supported(int F, int max_nodes) {
  if (F <= 0 || F % 4 != 0 || F > 256 || max_nodes <= 0) return 0;
  const int lpr = F <= 64 ? 16 : (F <= 128 ? 32 : 64);
  const int rw = 64 / lpr;
  const long long bytes = (long long)max_nodes * F * 4;
  const int groups = (max_nodes + rw - 1) / rw;
  return bytes <= SG_MAX_LDS && (groups + SG_WAVES - 1) / SG_WAVES <= SG_MAX_ITERS;
}

int isic_spmm_csr_graphs_f32(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, const float* bias,
                             float* out, int64_t n_rows, int F, float alpha, const float* addend, float addend_scale,
                             const int64_t* offsets, int n_graphs, int max_nodes, void* stream) {
  ISIC_CHECK_ARG(n_rows >= 0 && F > 0 && n_graphs >= 0 && max_nodes >= 0);
  if (n_rows == 0 || n_graphs == 0) return ISIC_OK;
  ISIC_CHECK_ARG(rowptr && col && val && x && out && offsets);
  const bool al16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias) |
                      reinterpret_cast<uintptr_t>(addend)) & 15) == 0;
  if (!al16 || !isic_spmm_csr_graphs_supported(F, max_nodes))      // the caller's generic path: same result, other schedule
    return isic_spmm_csr_f32(rowptr, col, val, x, bias, out, n_rows, F, alpha, addend, addend_scale, stream);
  SGArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val; a.x = x; a.bias = bias; a.out = out; a.offsets = offsets; a.addend = addend;
  a.F = F; a.max_nodes = max_nodes; a.alpha = alpha; a.addend_scale = addend_scale;
  const int lds = (int)((((long long)max_nodes * F * 4 + 1023) / 1024) * 1024);
  const int lpr = F <= 64 ? 16 : (F <= 128 ? 32 : 64);
  hipStream_t s = as_stream(stream);
  int rc;
  if (lpr == 16) rc = launch_sg<16>(a, n_graphs, lds, s);
  else if (lpr == 32) rc = launch_sg<32>(a, n_graphs, lds, s);
  else rc = launch_sg<64>(a, n_graphs, lds, s);
  return rc != ISIC_OK ? rc : isic_launch_status();
}

}  // extern "C"
