// Patch-graph kernels (gfx950): k-NN adjacency build, CSR-by-destination with GCN symmetric
// normalisation, and the neighbour gather / segmented sum (CSR SpMM) of message passing.
//
//   isic_knn_graph        03_build_graphs.py:37-54 / utils_g_mil.py:596-615 (distance GEMM + top-k)
//   isic_gcn_csr_build    PyG GCNConv.gcn_norm as called at 05_train_gnns.py:82,184-185
//   isic_spmm_csr_f32     GCNConv.propagate: out[dst] += w^ * x[src]   (05_train_gnns.py:184-185)
//
// The gather/scatter is re-expressed as a destination-major CSR so that the "scatter" becomes a
// segmented sum owned by one wave per destination row: no atomics, deterministic order (edges keep
// their edge_index order inside a row), every neighbour row is read with one coalesced wave-wide
// load, every output row written once.  The backward pass is the same kernel on the transposed CSR.
#include "common.h"

namespace {

// ================================================================= k-NN
// block = (graph, 16 query rows).  distances of the 16 queries to all N nodes of the graph via
// exact-fp32 MFMA (16x16x4) in chunks of 64 candidates (one 16x16 tile per wave), kept in LDS;
// then k rounds of wave-wide arg-min per query row (ties -> lower index).
__global__ void row_sqnorm_kernel(const float* __restrict__ x, int64_t T, int D, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= T) return;
  float s = 0.f;
  for (int j = lane; j < D; j += 64) { const float v = x[row * D + j]; s += v * v; }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ x, const float* __restrict__ sqn,
                                                   const int64_t* __restrict__ offsets, int D, int k, int max_nodes,
                                                   int64_t* __restrict__ nn_idx, float* __restrict__ nn_dist) {
  extern __shared__ __attribute__((aligned(16))) float dist[];   // [16][npad]
  const int g = blockIdx.y, rt = blockIdx.x;
  const int64_t lo = offsets[g];
  const int N = (int)(offsets[g + 1] - lo);
  const int r0 = rt * 16;
  if (r0 >= N) return;
  const int npad = ((max_nodes + 63) / 64) * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const bool vec = (D % 4) == 0;
  const int qrow = r0 + fr;                                   // this lane's query row (A operand)
  const float* qp = x + (lo + (qrow < N ? qrow : N - 1)) * D;
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int ccol = c0 + wave * 16 + fr;                     // this lane's candidate (B operand)
    const float* cp = x + (lo + (ccol < N ? ccol : N - 1)) * D;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < D; d0 += 16) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      const int d = d0 + 4 * fg;
      if (vec && d + 3 < D) {
        a = *reinterpret_cast<const float4*>(qp + d);
        b = *reinterpret_cast<const float4*>(cp + d);
      } else {
        if (d < D) { a.x = qp[d]; b.x = cp[d]; }
        if (d + 1 < D) { a.y = qp[d + 1]; b.y = cp[d + 1]; }
        if (d + 2 < D) { a.z = qp[d + 2]; b.z = cp[d + 2]; }
        if (d + 3 < D) { a.w = qp[d + 3]; b.w = cp[d + 3]; }
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
    // C map: row(query) = fg*4 + r, col(candidate) = fr
    const int col = c0 + wave * 16 + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = r0 + fg * 4 + r;
      float dv = INFINITY;
      if (q < N && col < N && q != col) {
        dv = (sqn[lo + q] + sqn[lo + col]) - 2.0f * acc[r];   // 03_build_graphs.py:47
        dv = fmaxf(dv, 0.f);                                  // :48 clamp(min=0)
      }
      dist[(fg * 4 + r) * npad + col] = dv;
    }
  }
  __syncthreads();
  // top-k: wave w owns query rows w*4 .. w*4+3
  for (int rr = 0; rr < 4; ++rr) {
    const int ql = wave * 4 + rr, q = r0 + ql;
    if (q >= N) continue;
    float* drow = dist + ql * npad;
    for (int j = 0; j < k; ++j) {
      float best = INFINITY;
      int bi = 0x7FFFFFFF;
      for (int c = lane; c < N; c += 64) {
        const float v = drow[c];
        if (v < best) { best = v; bi = c; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      const bool found = best < INFINITY;
      if (lane == 0) {
        nn_idx[(lo + q) * k + j] = found ? (int64_t)bi : (int64_t)-1;
        if (nn_dist) nn_dist[(lo + q) * k + j] = found ? best : INFINITY;
        if (found) drow[bi] = INFINITY;
      }
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
    }
  }
}

// ================================================================= GCN CSR
__global__ void csr_init_kernel(int* __restrict__ cnt, float* __restrict__ loopw, int64_t n) {
  // cnt holds 4 int arrays of n: cnt_in, cnt_out, fill_in, fill_out
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < 4 * n; i += stride) cnt[i] = 0;
  i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += stride) loopw[i] = 1.0f;
}

// mode 0 (GCN): self loops are dropped here and re-added (one per node) by the row kernels
// mode 1/2 (sum / mean aggregation, SAGE / GIN): every edge is kept as it is, no loop is added
__global__ void csr_count_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                 const float* __restrict__ w, int64_t E, int64_t n, int* __restrict__ cnt_in,
                                 int* __restrict__ cnt_out, float* __restrict__ loopw, int mode) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < E; e += stride) {
    const int64_t s = src[e], d = dst[e];
    if (mode == 0 && s == d) { loopw[s] = w ? w[e] : 1.0f; }
    else { atomicAdd(&cnt_in[d], 1); atomicAdd(&cnt_out[s], 1); }
  }
}

// single-block exclusive scan of (cnt[i] + extra) for both directions (extra = 1 slot for the GCN self loop)
__global__ __launch_bounds__(1024) void csr_scan_kernel(const int* __restrict__ cnt_in, const int* __restrict__ cnt_out,
                                                         int64_t n, int* __restrict__ rowptr, int* __restrict__ rowptr_t,
                                                         int extra) {
  __shared__ int part[2][1024];
  const int tid = threadIdx.x;
  const int64_t per = (n + 1023) / 1024;
  const int64_t b = tid * per, e = (b + per < n) ? b + per : n;
  int s0 = 0, s1 = 0;
  for (int64_t i = b; i < e; ++i) { s0 += cnt_in[i] + extra; s1 += cnt_out[i] + extra; }
  part[0][tid] = s0; part[1][tid] = s1;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v0 = 0, v1 = 0;
    if (tid >= off) { v0 = part[0][tid - off]; v1 = part[1][tid - off]; }
    __syncthreads();
    part[0][tid] += v0; part[1][tid] += v1;
    __syncthreads();
  }
  int r0 = part[0][tid] - s0, r1 = part[1][tid] - s1;   // exclusive prefix of this thread's chunk
  for (int64_t i = b; i < e; ++i) {
    rowptr[i] = r0; rowptr_t[i] = r1;
    r0 += cnt_in[i] + extra; r1 += cnt_out[i] + extra;
  }
  if (tid == 1023) { rowptr[n] = part[0][1023]; rowptr_t[n] = part[1][1023]; }
}

__global__ void csr_fill_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                const float* __restrict__ w, int64_t E, const int* __restrict__ rowptr,
                                const int* __restrict__ rowptr_t, int* __restrict__ fill_in, int* __restrict__ fill_out,
                                int* __restrict__ col, float* __restrict__ val, int* __restrict__ eid,
                                int* __restrict__ col_t, float* __restrict__ val_t, int* __restrict__ eid_t, int mode) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < E; e += stride) {
    const int64_t s = src[e], d = dst[e];
    if (mode == 0 && s == d) continue;
    const float wv = w ? w[e] : 1.0f;
    const int p = rowptr[d] + atomicAdd(&fill_in[d], 1);
    col[p] = (int)s; val[p] = wv; eid[p] = (int)e;
    const int pt = rowptr_t[s] + atomicAdd(&fill_out[s], 1);
    col_t[pt] = (int)d; val_t[pt] = wv; eid_t[pt] = (int)e;
  }
}

// per row: order entries by original edge id (deterministic), append the self loop, degree
__device__ void sort_row(int* col, float* val, int* eid, int b, int e) {
  for (int i = b + 1; i < e; ++i) {
    const int ke = eid[i], kc = col[i];
    const float kv = val[i];
    int j = i - 1;
    while (j >= b && eid[j] > ke) { eid[j + 1] = eid[j]; col[j + 1] = col[j]; val[j + 1] = val[j]; --j; }
    eid[j + 1] = ke; col[j + 1] = kc; val[j + 1] = kv;
  }
}
__global__ void csr_sort_kernel(const int* __restrict__ rowptr, const int* __restrict__ rowptr_t, int64_t n,
                                const float* __restrict__ loopw, int* __restrict__ col, float* __restrict__ val,
                                int* __restrict__ eid, int* __restrict__ col_t, float* __restrict__ val_t,
                                int* __restrict__ eid_t, float* __restrict__ dis, int mode) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 0) {
    {
      const int b = rowptr[i], e = rowptr[i + 1] - 1;   // last slot = self loop
      sort_row(col, val, eid, b, e);
      col[e] = (int)i; val[e] = loopw[i];
      float deg = 0.f;
      for (int p = b; p <= e; ++p) deg += val[p];        // deg[i] = sum of weights INTO i (+ loop)
      dis[i] = deg > 0.f ? 1.0f / sqrtf(deg) : 0.f;      // deg^-1/2, inf -> 0
    }
    {
      const int b = rowptr_t[i], e = rowptr_t[i + 1] - 1;
      sort_row(col_t, val_t, eid_t, b, e);
      col_t[e] = (int)i; val_t[e] = loopw[i];
    }
  } else {
    sort_row(col, val, eid, rowptr[i], rowptr[i + 1]);
    sort_row(col_t, val_t, eid_t, rowptr_t[i], rowptr_t[i + 1]);
    const int cnt = rowptr[i + 1] - rowptr[i];            // in-degree (number of incoming edges)
    dis[i] = (mode == 2 && cnt > 0) ? 1.0f / (float)cnt : (mode == 2 ? 0.f : 1.0f);
  }
}
__global__ void csr_norm_kernel(const int* __restrict__ rowptr, const int* __restrict__ rowptr_t, int64_t n,
                                const float* __restrict__ dis, const int* __restrict__ col, float* __restrict__ val,
                                const int* __restrict__ col_t, float* __restrict__ val_t, int mode) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 0) {
    // w^ = dis[src] * w * dis[dst]  (PyG order of operations)
    for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) val[p] = (dis[col[p]] * val[p]) * dis[i];        // row = dst
    for (int p = rowptr_t[i]; p < rowptr_t[i + 1]; ++p) val_t[p] = (dis[i] * val_t[p]) * dis[col_t[p]];  // row = src
  } else {
    // sum: w ; mean: w / in-degree(dst)   (dis[] holds 1 or 1/in-degree of the DESTINATION)
    for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) val[p] = val[p] * dis[i];
    for (int p = rowptr_t[i]; p < rowptr_t[i + 1]; ++p) val_t[p] = val_t[p] * dis[col_t[p]];
  }
}

// ================================================================= SpMM
// one wave per output row; lanes stride the F features (coalesced 256-B..1-KB row reads)
template <int VEC>
__global__ __launch_bounds__(256) void spmm_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                    const float* __restrict__ val, const float* __restrict__ x,
                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                    int64_t n_rows, int F, float alpha, const float* __restrict__ addend,
                                                    float addend_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int b = rowptr[row], e = rowptr[row + 1];
  for (int f0 = lane * VEC; f0 < F; f0 += 64 * VEC) {
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int p = b; p < e; ++p) {
      const float w = val[p];
      const float* xr = x + (int64_t)col[p] * F + f0;
      if (VEC == 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xr);
        acc[0] += w * xv.x; acc[1 % VEC] += w * xv.y; acc[2 % VEC] += w * xv.z; acc[3 % VEC] += w * xv.w;
      } else if (VEC == 2) {
        const float2 xv = *reinterpret_cast<const float2*>(xr);
        acc[0] += w * xv.x; acc[1 % VEC] += w * xv.y;
      } else {
        acc[0] += w * xr[0];
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float o = alpha * acc[v];
      if (bias) o += bias[f0 + v];
      if (addend) o += addend_scale * addend[row * F + f0 + v];
      out[row * F + f0 + v] = o;
    }
  }
}

// F <= 256, F % 4 == 0: a GROUP of LPR lanes (16 bytes of the row each) accumulates one run of edges; a block of 256 / LPR
// groups owns as many consecutive output rows.
// The one-wave-per-row kernel above walks its ~9 edges through two dependent loads each (column index, then the
// neighbour row) with nothing else in flight: latency-bound at a tenth of the HBM rate.  Here a group fetches the
// indices and weights of up to LPR edges with ONE coalesced load, broadcasts them by lane shuffles, and keeps four
// independent neighbour-row loads in flight.
// Round 3: HUB ROWS ARE SPLIT.  In a k-NN graph every row of one orientation has k + 1 entries, but the other
// orientation's rows are in-degrees -- 1 to 135 at k = 8 on 196 high-dimensional points (hubness), 5 % of the rows
// >= 32 -- and a launch was as long as its longest row's chain of dependent gathers (30 us against 16 us for the uniform
// orientation on the same bytes).  A block without a row of more than SPMM_SHORT entries works as before (one row per
// group).  Otherwise the block's rows are cut into ITEMS of <= `ch` (8, doubled until the block has <= MAXI items)
// consecutive entries, dealt round-robin to the groups; a row of one item is finished by its group, the partial sums of a
// split row go through LDS and are added IN ITEM ORDER by the row's own group: the result depends on the lengths of the
// block's rows only, never on timing (bit-reproducible), and a 135-entry row costs three rounds of two gathers.
constexpr int SPMM_THREADS = 256, UNR = 4;     // (12 gathers in flight measured slower: 0.25 vs 0.21 ms forward)
constexpr int SPMM_SHORT = 16, SPMM_CH_LOG2 = 3;
template <int LPR>
__global__ __launch_bounds__(SPMM_THREADS) void spmm_group_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                          const float* __restrict__ val, const float* __restrict__ x,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int64_t n_rows, int F, float alpha,
                                                          const float* __restrict__ addend, float addend_scale) {
  constexpr int G = SPMM_THREADS / LPR;                    // groups = rows per block
  constexpr int MAXI = LPR == 64 ? 16 : 32;                // items per block whose partial sums fit the LDS slots
  __shared__ float4 part[MAXI][LPR];
  const int lane = threadIdx.x & 63, gl = lane & (LPR - 1), g0 = lane - gl;      // lane in group, first lane of the group
  const int grp = threadIdx.x / LPR;
  const int64_t per_xcd = (gridDim.x + 7) / 8;
  const int64_t chunk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);      // may exceed the last chunk: rows >= n_rows
  const int64_t row0 = chunk * G;
  const bool fl = gl * 4 < F;                                                     // this lane holds features 4gl .. 4gl+3
  const float* xl = x + gl * 4;
  auto finish = [&](int64_t row, float4 acc) {
    if (!fl) return;
    float4 o = make_float4(alpha * acc.x, alpha * acc.y, alpha * acc.z, alpha * acc.w);
    if (bias) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + gl * 4);
      o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
    }
    if (addend) {
      const float4 av = *reinterpret_cast<const float4*>(addend + row * F + gl * 4);
      o.x += addend_scale * av.x; o.y += addend_scale * av.y; o.z += addend_scale * av.z; o.w += addend_scale * av.w;
    }
    *reinterpret_cast<float4*>(out + row * F + gl * 4) = o;
  };
  // sum of w * x[col] over the `deg` stored entries from slot b on, in slot order
  auto gather = [&](int b, int deg) -> float4 {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0;; base += LPR) {
      const int cnt = min(LPR, deg - base);                // entries of this group's run in this chunk (<= 0: none left)
      int maxcnt = cnt;                                    // wave-uniform trip count: the largest chunk of the wave's groups
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, o, 64));
      if (maxcnt <= 0) break;
      int c = 0;
      float w = 0.f;
      if (gl < cnt) { c = col[b + base + gl]; w = val[b + base + gl]; }
      for (int j = 0; j < maxcnt; j += UNR) {
        int cj[UNR];
        float wj[UNR];
        float4 xv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int src = g0 + ((j + u) & (LPR - 1));
          cj[u] = __shfl(c, src, 64);
          wj[u] = __shfl(w, src, 64);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j + u < cnt && fl) xv[u] = *reinterpret_cast<const float4*>(xl + (int64_t)cj[u] * F);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (j + u < cnt) {
            acc.x += wj[u] * xv[u].x; acc.y += wj[u] * xv[u].y; acc.z += wj[u] * xv[u].z; acc.w += wj[u] * xv[u].w;
          }
      }
    }
    return acc;
  };

  // the group's own row (rounds 1-2: the whole kernel) ...
  const int64_t row = row0 + grp;
  const bool row_ok = row < n_rows;
  const int b_own = row_ok ? rowptr[row] : 0, deg_own = row_ok ? rowptr[row + 1] - b_own : 0;
  // ... and the block's rows in the lanes of every wave (lane i < G: row i; all waves hold the same table)
  int rp_l = 0;
  if (lane <= G) {
    const int64_t r = row0 + lane;
    rp_l = rowptr[r < n_rows ? r : n_rows];
  }
  const int d_l = __shfl_down(rp_l, 1, 64) - rp_l;         // row length (lanes < G)
  const bool valid_l = lane < G && row0 + lane < n_rows;
  if (__ballot(valid_l && d_l > SPMM_SHORT) == 0) {        // block-uniform: no long row, one row per group
    const float4 acc = gather(b_own, deg_own);
    if (row_ok) finish(row, acc);
    return;
  }
  // item table: items of the row, exclusive prefix of the items; looked up by lane shuffles
  int sh = SPMM_CH_LOG2, it_l, incl, total;                // item length ch = 1 << sh = 8, 16, ...
  for (;;) {                                               // block-uniform
    it_l = valid_l ? max(1, (d_l + (1 << sh) - 1) >> sh) : 0;       // an empty row still has an item: its output is written
    incl = it_l;
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      incl += lane >= o ? up : 0;
    }
    total = __builtin_amdgcn_readlane(incl, G - 1);
    if (total <= MAXI) break;
    ++sh;
  }
  const int ofs_l = incl - it_l;
  const int ch = 1 << sh;
  for (int t0 = 0; t0 < total; t0 += G) {                  // block-uniform trip count
    const int t = t0 + grp;
    const bool item = t < total;
    int i = 0;                                             // the item's row: the last one with ofs <= t
#pragma unroll
    for (int k = 1; k < G; ++k) i += t >= __builtin_amdgcn_readlane(ofs_l, k) ? 1 : 0;     // (rows past n_rows: ofs = total > t)
    const int rb = __shfl(rp_l, i, 64), re = __shfl(rp_l, i + 1, 64);
    const int o_i = __shfl(ofs_l, i, 64), n_i = __shfl(it_l, i, 64);
    const int b = rb + (t - o_i) * ch;
    const float4 acc = gather(b, item ? min(ch, re - b) : 0);      // (0 entries: an empty row or no item)
    if (item) {
      if (n_i == 1) finish(row0 + i, acc);
      else part[t][gl] = acc;
    }
  }
  __syncthreads();
  const int o_g = __shfl(ofs_l, grp, 64), n_g = __shfl(it_l, grp, 64);
  if (n_g > 1) {                                           // a split row: its items in order
    float4 acc = part[o_g][gl];
    for (int c = 1; c < n_g; ++c) {
      const float4 p = part[o_g + c][gl];
      acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    finish(row, acc);
  }
}

// perm_t[pt] = CSR slot of the edge held by transposed slot pt (self-loop slots map to each other)
__global__ void csr_pos_kernel(const int* __restrict__ rowptr, int64_t n, const int* __restrict__ eid,
                               int* __restrict__ pos_of_edge, int mode) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int e = rowptr[i + 1] - (mode == 0 ? 1 : 0);
  for (int p = rowptr[i]; p < e; ++p) pos_of_edge[eid[p]] = p;
}
__global__ void csr_perm_kernel(const int* __restrict__ rowptr, const int* __restrict__ rowptr_t, int64_t n,
                                const int* __restrict__ eid_t, const int* __restrict__ pos_of_edge,
                                int* __restrict__ perm_t, int mode) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int e = rowptr_t[i + 1] - (mode == 0 ? 1 : 0);
  for (int pt = rowptr_t[i]; pt < e; ++pt) perm_t[pt] = pos_of_edge[eid_t[pt]];
  if (mode == 0) perm_t[e] = rowptr[i + 1] - 1;
}

inline int grid_for(int64_t n, int block, int cap = 4096) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}


// One launch assembles a step's batched CSR (and its transpose) out of per-graph CSR pieces built once
// (train.GraphStore): B graphs picked by `sel`, every graph with n nodes and nnz stored entries, local ids.
//   rowptr_out[b*n + r] = rowptr[g*n + r] + b*nnz (+ the closing entry B*nnz),  col_out[b*nnz + e] = col[g*nnz + e] + b*n,
//   perm_out[b*nnz + e] = perm[g*nnz + e] + b*nnz,  val copied;  g = sel[b].
// Replaces seven index_selects, five integer adds, two concatenations and their helper kernels per optimizer step
// (05_train_gnns.py:340-343 re-uploads x and edge_index of every graph at every step instead).
__global__ __launch_bounds__(256) void csr_batch_assemble_kernel(
    const int64_t* __restrict__ sel, int B, int n, int nnz, const int* __restrict__ rowptr, const int* __restrict__ rowptr_t,
    const int* __restrict__ col, const int* __restrict__ col_t, const float* __restrict__ val, const float* __restrict__ val_t,
    const int* __restrict__ perm_t, int* __restrict__ rowptr_o, int* __restrict__ rowptr_t_o, int* __restrict__ col_o,
    int* __restrict__ col_t_o, float* __restrict__ val_o, float* __restrict__ val_t_o, int* __restrict__ perm_t_o,
    int* __restrict__ rowidx_o) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ne = (int64_t)B * nnz, nr = (int64_t)B * n;
  if (i < ne) {
    const int b = (int)(i / nnz), e = (int)(i - (int64_t)b * nnz);
    const int64_t src = sel[b] * nnz + e;
    col_o[i] = col[src] + b * n;
    col_t_o[i] = col_t[src] + b * n;
    val_o[i] = val[src];
    val_t_o[i] = val_t[src];
    perm_t_o[i] = perm_t[src] + b * nnz;
  }
  if (i < nr) {
    const int b = (int)(i / n), r = (int)(i - (int64_t)b * n);
    const int64_t src = sel[b] * n + r;
    rowptr_o[i] = rowptr[src] + b * nnz;
    rowptr_t_o[i] = rowptr_t[src] + b * nnz;
    if (rowidx_o) rowidx_o[i] = (int)src;                  // node row b*n + r of the batch = row sel[b]*n + r of the record store
  } else if (i == nr) {
    rowptr_o[i] = (int)ne;
    rowptr_t_o[i] = (int)ne;
  }
}

}  // namespace

bool isic_knn_gram_supported(int D, int k, int max_nodes);
int isic_knn_gram_launch(const float* x, const int64_t* offsets, int G, int D, int k, int64_t* nn_idx, float* nn_dist,
                         hipStream_t stream);

extern "C" {

int isic_knn_graph(const float* x, const int64_t* offsets, int G, int D, int k, int max_nodes, int64_t total_nodes,
                   int64_t* nn_idx, float* nn_dist, float* workspace_sqnorm, void* stream) {
  ISIC_CHECK_ARG(G >= 0 && D > 0 && k > 0 && max_nodes >= 0 && total_nodes >= 0);
  if (G == 0 || total_nodes == 0 || max_nodes == 0) return ISIC_OK;
  ISIC_CHECK_ARG(x && offsets && nn_idx && workspace_sqnorm);
  // graphs of <= 208 nodes: the whole Gram matrix per workgroup, top-k out of the accumulators (knn_gram.hip)
  if (isic_knn_gram_supported(D, k, max_nodes)) {
    const int rc = isic_knn_gram_launch(x, offsets, G, D, k, nn_idx, nn_dist, as_stream(stream));
    return rc != ISIC_OK ? rc : isic_launch_status();
  }
  const int npad = ((max_nodes + 63) / 64) * 64;
  const size_t lds = (size_t)16 * npad * sizeof(float);
  if (lds > 150 * 1024) return ISIC_ERR_UNSUPPORTED;   // > ~2400 nodes per graph
  if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(knn_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return ISIC_ERR_LAUNCH;
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((total_nodes + 3) / 4)), dim3(256), 0, as_stream(stream), x,
                     total_nodes, D, workspace_sqnorm);
  dim3 grid(ceil_div(max_nodes, 16), G);
  ISIC_CHECK_ARG(grid.y <= 65535u);
  hipLaunchKernelGGL(knn_kernel, grid, dim3(256), lds, as_stream(stream), x, workspace_sqnorm, offsets, D, k, max_nodes,
                     nn_idx, nn_dist);
  return isic_launch_status();
}

size_t isic_gcn_csr_workspace_bytes(int64_t n_nodes, int64_t E) {
  // cnt_in, cnt_out, fill_in, fill_out (int) | loopw, dis (float) | eid, eid_t (int, E+n each) | pos_of_edge (int, E)
  return (size_t)(6 * n_nodes + 2 * (E + n_nodes) + E + 64) * 4;
}

int isic_gcn_csr_build(const int64_t* src, const int64_t* dst, const float* edge_weight, int64_t E, int64_t n_nodes,
                       int mode, int32_t* rowptr, int32_t* col, float* val, int32_t* rowptr_t, int32_t* col_t,
                       float* val_t, int32_t* perm_t, void* workspace, size_t workspace_bytes, void* stream) {
  ISIC_CHECK_ARG(E >= 0 && n_nodes > 0 && rowptr && col && val && rowptr_t && col_t && val_t && workspace);
  ISIC_CHECK_ARG(mode >= 0 && mode <= 2);
  ISIC_CHECK_ARG(E == 0 || (src && dst));
  if (E + n_nodes > 0x7FFFFFF0LL) return ISIC_ERR_UNSUPPORTED;
  if (workspace_bytes < isic_gcn_csr_workspace_bytes(n_nodes, E)) return ISIC_ERR_WORKSPACE;
  int* cnt = reinterpret_cast<int*>(workspace);
  int *cnt_in = cnt, *cnt_out = cnt + n_nodes, *fill_in = cnt + 2 * n_nodes, *fill_out = cnt + 3 * n_nodes;
  float* loopw = reinterpret_cast<float*>(cnt + 4 * n_nodes);
  float* dis = loopw + n_nodes;
  int* eid = reinterpret_cast<int*>(dis + n_nodes);
  int* eid_t = eid + (E + n_nodes);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(csr_init_kernel, dim3(grid_for(4 * n_nodes, 256)), dim3(256), 0, s, cnt, loopw, n_nodes);
  if (E > 0)
    hipLaunchKernelGGL(csr_count_kernel, dim3(grid_for(E, 256)), dim3(256), 0, s, src, dst, edge_weight, E, n_nodes,
                       cnt_in, cnt_out, loopw, mode);
  hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, s, cnt_in, cnt_out, n_nodes, rowptr, rowptr_t,
                     mode == 0 ? 1 : 0);
  if (E > 0)
    hipLaunchKernelGGL(csr_fill_kernel, dim3(grid_for(E, 256)), dim3(256), 0, s, src, dst, edge_weight, E, rowptr,
                       rowptr_t, fill_in, fill_out, col, val, eid, col_t, val_t, eid_t, mode);
  const int g = (int)((n_nodes + 255) / 256);
  hipLaunchKernelGGL(csr_sort_kernel, dim3(g), dim3(256), 0, s, rowptr, rowptr_t, n_nodes, loopw, col, val, eid, col_t,
                     val_t, eid_t, dis, mode);
  hipLaunchKernelGGL(csr_norm_kernel, dim3(g), dim3(256), 0, s, rowptr, rowptr_t, n_nodes, dis, col, val, col_t, val_t,
                     mode);
  if (perm_t) {
    int* pos_of_edge = eid_t + (E + n_nodes);
    hipLaunchKernelGGL(csr_pos_kernel, dim3(g), dim3(256), 0, s, rowptr, n_nodes, eid, pos_of_edge, mode);
    hipLaunchKernelGGL(csr_perm_kernel, dim3(g), dim3(256), 0, s, rowptr, rowptr_t, n_nodes, eid_t, pos_of_edge, perm_t,
                       mode);
  }
  return isic_launch_status();
}

int isic_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, const float* bias,
                      float* out, int64_t n_rows, int F, float alpha, const float* addend, float addend_scale,
                      void* stream) {
  ISIC_CHECK_ARG(n_rows >= 0 && F > 0);
  if (n_rows == 0) return ISIC_OK;
  ISIC_CHECK_ARG(rowptr && col && val && x && out);
  const dim3 grid((unsigned)((n_rows + 3) / 4));
  hipStream_t s = as_stream(stream);
  const bool al16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  const bool al16_all = al16 && ((reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(addend)) & 15) == 0;
  if (F % 4 == 0 && F <= 256 && al16_all) {
    const int lpr = F <= 64 ? 16 : (F <= 128 ? 32 : 64);
    const int rows_per_block = SPMM_THREADS / lpr;
    const dim3 g2((unsigned)((((n_rows + rows_per_block - 1) / rows_per_block) + 7) / 8 * 8));   // multiple of 8: bijective XCD remap
    if (lpr == 16) hipLaunchKernelGGL(spmm_group_kernel<16>, g2, dim3(SPMM_THREADS), 0, s, rowptr, col, val, x, bias, out, n_rows, F, alpha, addend, addend_scale);
    else if (lpr == 32) hipLaunchKernelGGL(spmm_group_kernel<32>, g2, dim3(SPMM_THREADS), 0, s, rowptr, col, val, x, bias, out, n_rows, F, alpha, addend, addend_scale);
    else hipLaunchKernelGGL(spmm_group_kernel<64>, g2, dim3(SPMM_THREADS), 0, s, rowptr, col, val, x, bias, out, n_rows, F, alpha, addend, addend_scale);
    return isic_launch_status();
  }
  if (F % 4 == 0 && al16 && F >= 256)
    hipLaunchKernelGGL(spmm_kernel<4>, grid, dim3(256), 0, s, rowptr, col, val, x, bias, out, n_rows, F, alpha, addend, addend_scale);
  else if (F % 2 == 0 && al16 && F >= 128)
    hipLaunchKernelGGL(spmm_kernel<2>, grid, dim3(256), 0, s, rowptr, col, val, x, bias, out, n_rows, F, alpha, addend, addend_scale);
  else
    hipLaunchKernelGGL(spmm_kernel<1>, grid, dim3(256), 0, s, rowptr, col, val, x, bias, out, n_rows, F, alpha, addend, addend_scale);
  return isic_launch_status();
}

int isic_csr_batch_assemble(const int64_t* sel, int B, int n, int nnz, const int32_t* rowptr, const int32_t* rowptr_t,
                            const int32_t* col, const int32_t* col_t, const float* val, const float* val_t,
                            const int32_t* perm_t, int32_t* rowptr_out, int32_t* rowptr_t_out, int32_t* col_out,
                            int32_t* col_t_out, float* val_out, float* val_t_out, int32_t* perm_t_out, int32_t* row_index_out,
                            void* stream) {
  ISIC_CHECK_ARG(B >= 0 && n > 0 && nnz >= 0);
  if (B == 0) return ISIC_OK;
  ISIC_CHECK_ARG(sel && rowptr && rowptr_t && col && col_t && val && val_t && perm_t && rowptr_out && rowptr_t_out && col_out &&
                 col_t_out && val_out && val_t_out && perm_t_out);
  ISIC_CHECK_ARG((int64_t)B * nnz < 0x7FFFFFFFLL && (int64_t)B * n < 0x7FFFFFFFLL);
  const int64_t ne = (int64_t)B * nnz, nr = (int64_t)B * n + 1;
  const int64_t work = ne > nr ? ne : nr;
  hipLaunchKernelGGL(csr_batch_assemble_kernel, dim3((unsigned)ceil_div64(work, 256)), dim3(256), 0, as_stream(stream), sel, B, n,
                     nnz, rowptr, rowptr_t, col, col_t, val, val_t, perm_t, rowptr_out, rowptr_t_out, col_out, col_t_out,
                     val_out, val_t_out, perm_t_out, row_index_out);
  return isic_launch_status();
}

}  // extern "C"
