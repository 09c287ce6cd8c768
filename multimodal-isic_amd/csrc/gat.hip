// Graph attention (GATConv) message passing on the destination-major CSR (gfx950).
//
// PyG ``GATConv(in, F, heads=H, concat=True, dropout=p)`` as called at 05_train_gnns.py:83-86
// (the tuned graph model of hypermarameters.yml:121-141): x' = lin(x) viewed [N,H,F];
// al[n,h] = <x'[n,h,:], att_src[h,:]>, ar[n,h] = <x'[n,h,:], att_dst[h,:]>; self loops re-added;
// e = leaky_relu(al[src] + ar[dst], 0.2); alpha = softmax over the edges INTO dst (per head);
// dropout on alpha; out[dst,h,:] = sum alpha * x'[src,h,:] (+ bias).
//
// The per-destination edge softmax is exactly the segmented softmax the CSR gives for free: one
// wave owns one destination row, walks its (short) edge list twice from registers / L2 -- scores,
// then an online-softmax weighted sum of coalesced neighbour rows -- no atomics, no scatter.  The
// backward pass is two such sweeps: by destination (d alpha -> d e, d ar) and by source through the
// transposed CSR (d x', d al), linked by the edge permutation perm_t the CSR build emits.
#include "common.h"

namespace {

constexpr int MAXH = 8;

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

// al[n,h], ar[n,h]: one wave per node
__global__ __launch_bounds__(256) void gat_scores_kernel(const float* __restrict__ xp, const float* __restrict__ att_src,
                                                          const float* __restrict__ att_dst, float* __restrict__ al,
                                                          float* __restrict__ ar, int64_t N, int H, int F) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  for (int h = 0; h < H; ++h) {
    float s = 0.f, d = 0.f;
    for (int f = lane; f < F; f += 64) {
      const float v = xp[(n * H + h) * F + f];
      s += v * att_src[h * F + f];
      d += v * att_dst[h * F + f];
    }
    s = wave_sum(s); d = wave_sum(d);
    if (lane == 0) { al[n * H + h] = s; ar[n * H + h] = d; }
  }
}

// forward: wave per destination row
__global__ __launch_bounds__(256) void gat_fwd_kernel(const float* __restrict__ xp, const float* __restrict__ al,
                                                       const float* __restrict__ ar, const int* __restrict__ rowptr,
                                                       const int* __restrict__ col, const float* __restrict__ bias,
                                                       float* __restrict__ out, float* __restrict__ alpha, int64_t N,
                                                       int H, int F, float slope, unsigned thr, float scale,
                                                       unsigned long long seed, unsigned long long stream_id) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const int b = rowptr[i], e = rowptr[i + 1];
  for (int h = 0; h < H; ++h) {
    const float ari = ar[i * H + h];
    // pass 1 (lanes over edges): row max and sum of exp
    float m = -INFINITY;
    for (int p = b + lane; p < e; p += 64) m = fmaxf(m, lrelu(al[(int64_t)col[p] * H + h] + ari, slope));
    m = wave_max(m);
    float s = 0.f;
    for (int p = b + lane; p < e; p += 64) s += expf(lrelu(al[(int64_t)col[p] * H + h] + ari, slope) - m);
    s = wave_sum(s);
    // pass 2 (lanes over features): weighted sum of neighbour rows
    for (int f0 = lane; f0 < F; f0 += 64) {
      float acc = 0.f;
      for (int p = b; p < e; ++p) {
        const int src = col[p];
        float a = expf(lrelu(al[(int64_t)src * H + h] + ari, slope) - m) / s;
        if (f0 == lane && lane == 0) alpha[(int64_t)p * H + h] = a;      // pre-dropout alpha, once per edge
        if (thr) a = philox_word((unsigned long long)p * H + h, seed, stream_id) >= thr ? a * scale : 0.f;
        acc += a * xp[((int64_t)src * H + h) * F + f0];
      }
      out[(i * H + h) * F + f0] = acc + (bias ? bias[h * F + f0] : 0.f);
    }
  }
}

// backward sweep 1, wave per destination row: d e (per edge, per head) and d ar
__global__ __launch_bounds__(256) void gat_bwd_dst_kernel(const float* __restrict__ dout, const float* __restrict__ xp,
                                                           const float* __restrict__ alpha, const float* __restrict__ al,
                                                           const float* __restrict__ ar, const int* __restrict__ rowptr,
                                                           const int* __restrict__ col, float* __restrict__ de,
                                                           float* __restrict__ dar, int64_t N, int H, int F, float slope,
                                                           unsigned thr, float scale, unsigned long long seed,
                                                           unsigned long long stream_id) {
  constexpr int MAXE = 512;                       // edges of one row kept in LDS (longer rows go through global memory)
  __shared__ float sda[4][MAXE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= N) return;
  const int b = rowptr[i], e = rowptr[i + 1];
  const bool in_lds = (e - b) <= MAXE;
  for (int h = 0; h < H; ++h) {
    const float* drow = dout + (i * H + h) * F;
    // d alpha_p = m_p * <dout[i,h,:], x'[src_p,h,:]> ; parked per edge, and sum_k alpha_k d alpha_k accumulated
    float dot = 0.f;
    for (int p = b; p < e; ++p) {
      const float* xr = xp + ((int64_t)col[p] * H + h) * F;
      float d = 0.f;
      for (int f = lane; f < F; f += 64) d += drow[f] * xr[f];
      d = wave_sum(d);
      if (thr) d = philox_word((unsigned long long)p * H + h, seed, stream_id) >= thr ? d * scale : 0.f;
      dot += alpha[(int64_t)p * H + h] * d;
      if (lane == 0) {
        if (in_lds) sda[wave][p - b] = d;
        else de[(int64_t)p * H + h] = d;
      }
    }
    if (in_lds) __builtin_amdgcn_wave_barrier();
    else __threadfence();                          // lane 0's global stores must be visible to the other lanes
    const float ari = ar[i * H + h];
    float sum_de = 0.f;
    for (int p = b + lane; p < e; p += 64) {
      const float a = alpha[(int64_t)p * H + h];
      const float pre = al[(int64_t)col[p] * H + h] + ari;
      const float da = in_lds ? sda[wave][p - b] : __builtin_nontemporal_load(&de[(int64_t)p * H + h]);
      const float g = a * (da - dot) * (pre > 0.f ? 1.f : slope);
      de[(int64_t)p * H + h] = g;
      sum_de += g;
    }
    sum_de = wave_sum(sum_de);
    if (lane == 0) dar[i * H + h] = sum_de;
    __builtin_amdgcn_wave_barrier();               // sda is reused by the next head
  }
}

// backward sweep 2, wave per source row (transposed CSR): d al and d x'
__global__ __launch_bounds__(256) void gat_bwd_src_kernel(const float* __restrict__ dout, const float* __restrict__ alpha,
                                                           const float* __restrict__ de, const float* __restrict__ dar,
                                                           const int* __restrict__ rowptr_t, const int* __restrict__ col_t,
                                                           const int* __restrict__ perm_t, const float* __restrict__ att_src,
                                                           const float* __restrict__ att_dst, float* __restrict__ dxp,
                                                           float* __restrict__ dal, int64_t N, int H, int F, unsigned thr,
                                                           float scale, unsigned long long seed,
                                                           unsigned long long stream_id) {
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= N) return;
  const int b = rowptr_t[s], e = rowptr_t[s + 1];
  for (int h = 0; h < H; ++h) {
    float g = 0.f;
    for (int pt = b + lane; pt < e; pt += 64) g += de[(int64_t)perm_t[pt] * H + h];
    g = wave_sum(g);
    if (lane == 0) dal[s * H + h] = g;
    const float gr = dar[s * H + h];
    for (int f0 = lane; f0 < F; f0 += 64) {
      float acc = g * att_src[h * F + f0] + gr * att_dst[h * F + f0];
      for (int pt = b; pt < e; ++pt) {
        const int p = perm_t[pt];
        float a = alpha[(int64_t)p * H + h];
        if (thr) a = philox_word((unsigned long long)p * H + h, seed, stream_id) >= thr ? a * scale : 0.f;
        acc += a * dout[((int64_t)col_t[pt] * H + h) * F + f0];
      }
      dxp[(s * H + h) * F + f0] = acc;
    }
  }
}

}  // namespace

extern "C" {

int isic_gat_scores(const float* xp, const float* att_src, const float* att_dst, float* al, float* ar, int64_t N, int H,
                    int F, void* stream) {
  ISIC_CHECK_ARG(N >= 0 && H > 0 && F > 0);
  if (N == 0) return ISIC_OK;
  ISIC_CHECK_ARG(xp && att_src && att_dst && al && ar);
  hipLaunchKernelGGL(gat_scores_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, as_stream(stream), xp, att_src,
                     att_dst, al, ar, N, H, F);
  return isic_launch_status();
}

int isic_gat_fwd(const float* xp, const float* al, const float* ar, const int32_t* rowptr, const int32_t* col,
                 const float* bias, float* out, float* alpha, int64_t N, int H, int F, float negative_slope,
                 uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id, void* stream) {
  ISIC_CHECK_ARG(N >= 0 && H > 0 && F > 0);
  if (N == 0) return ISIC_OK;
  ISIC_CHECK_ARG(xp && al && ar && rowptr && col && out && alpha);
  hipLaunchKernelGGL(gat_fwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, as_stream(stream), xp, al, ar, rowptr,
                     col, bias, out, alpha, N, H, F, negative_slope, drop_threshold, drop_scale,
                     (unsigned long long)seed, (unsigned long long)stream_id);
  return isic_launch_status();
}

int isic_gat_bwd(const float* dout, const float* xp, const float* alpha, const float* al, const float* ar,
                 const float* att_src, const float* att_dst, const int32_t* rowptr, const int32_t* col,
                 const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t, float* de, float* dar, float* dal,
                 float* dxp, int64_t N, int H, int F, float negative_slope, uint32_t drop_threshold, float drop_scale,
                 uint64_t seed, uint64_t stream_id, void* stream) {
  ISIC_CHECK_ARG(N >= 0 && H > 0 && F > 0);
  if (N == 0) return ISIC_OK;
  ISIC_CHECK_ARG(dout && xp && alpha && al && ar && att_src && att_dst && rowptr && col && rowptr_t && col_t && perm_t &&
                 de && dar && dal && dxp);
  const dim3 grid((unsigned)((N + 3) / 4));
  hipLaunchKernelGGL(gat_bwd_dst_kernel, grid, dim3(256), 0, as_stream(stream), dout, xp, alpha, al, ar, rowptr, col, de,
                     dar, N, H, F, negative_slope, drop_threshold, drop_scale, (unsigned long long)seed,
                     (unsigned long long)stream_id);
  hipLaunchKernelGGL(gat_bwd_src_kernel, grid, dim3(256), 0, as_stream(stream), dout, alpha, de, dar, rowptr_t, col_t,
                     perm_t, att_src, att_dst, dxp, dal, N, H, F, drop_threshold, drop_scale, (unsigned long long)seed,
                     (unsigned long long)stream_id);
  return isic_launch_status();
}

}  // extern "C"
