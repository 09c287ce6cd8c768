// Edge-attention message passing on the destination-major CSR (gfx950): the remaining PyG layers the reference's
// GraphMIL can select (05_train_gnns.py:94-106) -- GATv2Conv, TransformerConv(beta=True) and FAConv.
//
//   GATv2Conv       logit[e,h] = sum_f att[h,f] * leaky_relu(xl[src,h,f] + xr[dst,h,f]); alpha = softmax over the edges
//                   INTO dst; out[dst,h,:] = sum alpha * xl[src,h,:]                       (self loops re-added, 'gcn' CSR)
//   TransformerConv logit[e,h] = <q[dst,h,:], k[src,h,:]> / sqrt(F); alpha = softmax over the edges into dst;
//                   out[dst,h,:] = sum alpha * v[src,h,:]                                  (edges as given, 'sum' CSR)
//   FAConv          out[dst,:] = sum tanh(al[src] + ar[dst]) * w[e] * x[src,:] + eps * x0[dst,:], w = GCN normalisation
//
// Same plan as gat.hip: one wave owns one destination row, walks its short edge list -- per-edge logits by a
// lane-parallel dot product over the gathered rows, an online softmax, then the weighted sum of coalesced neighbour
// rows -- no atomics, no scatter.  Backward is a destination sweep (d logit per edge, the destination-side gradient)
// and a source sweep over the transposed CSR (source-side and value gradients), linked by perm_t.  Only the tiny
// d att[H,F] of GATv2 is reduced across rows: persistent blocks keep it in registers and add it once.
#include "common.h"

namespace {

constexpr int EA_GATV2 = 0, EA_DOT = 1;
constexpr int MAXE = 512;                        // edges of one row kept in LDS (longer rows go through global memory)

__device__ __forceinline__ float ea_lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

// per-edge logit of edge (src -> i), head h: every lane returns the same value
template <int MODE>
__device__ __forceinline__ float edge_logit(const float* __restrict__ ks_row, const float* __restrict__ qd_row,
                                            const float* __restrict__ att_h, int F, float slope, float scale, int lane) {
  float d = 0.f;
  for (int f = lane; f < F; f += 64) {
    if (MODE == EA_GATV2) d += att_h[f] * ea_lrelu(ks_row[f] + qd_row[f], slope);
    else d += qd_row[f] * ks_row[f];
  }
  d = wave_sum(d);
  return MODE == EA_DOT ? d * scale : d;
}

// forward: wave per destination row.  alpha[nnz,H] receives the PRE-dropout attention.
template <int MODE>
__global__ __launch_bounds__(256) void edge_attn_fwd_kernel(const float* __restrict__ ks, const float* __restrict__ qd,
                                                             const float* __restrict__ v, const float* __restrict__ att,
                                                             const int* __restrict__ rowptr, const int* __restrict__ col,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             float* __restrict__ alpha, int64_t N, int H, int F,
                                                             float slope, float scale, unsigned thr, float dscale,
                                                             unsigned long long seed, unsigned long long stream_id) {
  __shared__ float slog[4][MAXE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= N) return;
  const int b = rowptr[i], e = rowptr[i + 1];
  const bool in_lds = (e - b) <= MAXE;
  for (int h = 0; h < H; ++h) {
    const float* qrow = qd + (i * H + h) * F;
    const float* att_h = att ? att + h * F : nullptr;
    float m = -INFINITY, s = 0.f;                                  // online softmax over the row's edges
    for (int p = b; p < e; ++p) {
      const float lg = edge_logit<MODE>(ks + ((int64_t)col[p] * H + h) * F, qrow, att_h, F, slope, scale, lane);
      if (lane == 0) {
        if (in_lds) slog[wave][p - b] = lg;
        else alpha[(int64_t)p * H + h] = lg;
      }
      const float mn = fmaxf(m, lg);
      s = s * expf(m - mn) + expf(lg - mn);
      m = mn;
    }
    if (in_lds) __builtin_amdgcn_wave_barrier();
    else __threadfence();
    const float inv = e > b ? 1.f / s : 0.f;
    for (int f0 = lane; f0 < F; f0 += 64) {
      float acc = 0.f;
      for (int p = b; p < e; ++p) {
        const float lg = in_lds ? slog[wave][p - b] : __builtin_nontemporal_load(&alpha[(int64_t)p * H + h]);
        float a = expf(lg - m) * inv;
        if (thr) a = philox_word((unsigned long long)p * H + h, seed, stream_id) >= thr ? a * dscale : 0.f;
        acc += a * v[((int64_t)col[p] * H + h) * F + f0];
      }
      out[(i * H + h) * F + f0] = acc + (bias ? bias[h * F + f0] : 0.f);
    }
    __builtin_amdgcn_wave_barrier();                               // every lane has read the logits
    for (int p = b + lane; p < e; p += 64) {
      const float lg = in_lds ? slog[wave][p - b] : alpha[(int64_t)p * H + h];
      alpha[(int64_t)p * H + h] = expf(lg - m) * inv;
    }
    __builtin_amdgcn_wave_barrier();                               // slog is reused by the next head
  }
}

// backward sweep 1: persistent blocks, wave per destination row.  de[nnz,H] = d logit; dqd[N,H,F] = destination-side
// gradient (d q for DOT, d xr for GATv2); datt[H,F] (GATv2) accumulated in registers over all rows of the wave.
template <int MODE>
__global__ __launch_bounds__(256) void edge_attn_bwd_dst_kernel(const float* __restrict__ dout, const float* __restrict__ ks,
                                                                 const float* __restrict__ qd, const float* __restrict__ v,
                                                                 const float* __restrict__ att, const float* __restrict__ alpha,
                                                                 const int* __restrict__ rowptr, const int* __restrict__ col,
                                                                 float* __restrict__ de, float* __restrict__ dqd,
                                                                 float* __restrict__ datt, int64_t N, int H, int F, float slope,
                                                                 float scale, unsigned thr, float dscale,
                                                                 unsigned long long seed, unsigned long long stream_id) {
  __shared__ float sda[4][MAXE];
  // d att[h, f]: lane l owns the elements f = l + 64 q of every head, slot = h * QF + q with QF = ceil(F / 64).  Up to
  // MAXV slots stay in registers over all rows of the wave (any F, any H with H * QF <= MAXV: 4 heads x 256, 4 x 32,
  // 1 x 1024 ...); wider layers add their row's contribution with one atomic per element instead.
  constexpr int MAXV = 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float datt_acc[MAXV];
#pragma unroll
  for (int q = 0; q < MAXV; ++q) datt_acc[q] = 0.f;
  const int QF = (F + 63) / 64;
  const bool in_regs = H * QF <= MAXV;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < N; i += (int64_t)gridDim.x * 4) {
    const int b = rowptr[i], e = rowptr[i + 1];
    const bool in_lds = (e - b) <= MAXE;
    for (int h = 0; h < H; ++h) {
      const float* drow = dout + (i * H + h) * F;
      float dot = 0.f;
      for (int p = b; p < e; ++p) {
        const float* vr = v + ((int64_t)col[p] * H + h) * F;
        float d = 0.f;
        for (int f = lane; f < F; f += 64) d += drow[f] * vr[f];
        d = wave_sum(d);
        if (thr) d = philox_word((unsigned long long)p * H + h, seed, stream_id) >= thr ? d * dscale : 0.f;
        dot += alpha[(int64_t)p * H + h] * d;
        if (lane == 0) {
          if (in_lds) sda[wave][p - b] = d;
          else de[(int64_t)p * H + h] = d;
        }
      }
      if (in_lds) __builtin_amdgcn_wave_barrier();
      else __threadfence();
      for (int p = b + lane; p < e; p += 64) {
        const float da = in_lds ? sda[wave][p - b] : __builtin_nontemporal_load(&de[(int64_t)p * H + h]);
        de[(int64_t)p * H + h] = alpha[(int64_t)p * H + h] * (da - dot);      // softmax backward
      }
      __threadfence();                                             // d logit of this row is re-read just below
      const float* qrow = qd + (i * H + h) * F;
      for (int f0 = lane, q = 0; f0 < F; f0 += 64, ++q) {
        float acc = 0.f, da_acc = 0.f;
        for (int p = b; p < e; ++p) {
          const float g = __builtin_nontemporal_load(&de[(int64_t)p * H + h]);
          const float kv = ks[((int64_t)col[p] * H + h) * F + f0];
          if (MODE == EA_GATV2) {
            const float sv = kv + qrow[f0];
            acc += g * att[h * F + f0] * (sv > 0.f ? 1.f : slope);
            da_acc += g * ea_lrelu(sv, slope);
          } else {
            acc += g * kv * scale;
          }
        }
        dqd[(i * H + h) * F + f0] = acc;
        if (MODE == EA_GATV2) {
          if (in_regs) {
            const int slot = h * QF + q;
#pragma unroll
            for (int t = 0; t < MAXV; ++t) datt_acc[t] += (t == slot) ? da_acc : 0.f;
          } else {
            atomicAdd(datt + h * F + f0, da_acc);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();                             // sda is reused by the next head
    }
  }
  if (MODE == EA_GATV2 && in_regs) {
#pragma unroll
    for (int t = 0; t < MAXV; ++t) {
      const int h = t / QF, f = lane + 64 * (t % QF);
      if (h < H && f < F) atomicAdd(datt + h * F + f, datt_acc[t]);
    }
  }
}

// backward sweep 2, wave per source row (transposed CSR): dks[N,H,F] (source-side logit gradient; for GATv2 the value
// gradient is added, v == xl) and dv[N,H,F] (DOT only).
template <int MODE>
__global__ __launch_bounds__(256) void edge_attn_bwd_src_kernel(const float* __restrict__ dout, const float* __restrict__ ks,
                                                                 const float* __restrict__ qd, const float* __restrict__ att,
                                                                 const float* __restrict__ alpha, const float* __restrict__ de,
                                                                 const int* __restrict__ rowptr_t, const int* __restrict__ col_t,
                                                                 const int* __restrict__ perm_t, float* __restrict__ dks,
                                                                 float* __restrict__ dv, int64_t N, int H, int F, float slope,
                                                                 float scale, unsigned thr, float dscale,
                                                                 unsigned long long seed, unsigned long long stream_id) {
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= N) return;
  const int b = rowptr_t[s], e = rowptr_t[s + 1];
  for (int h = 0; h < H; ++h) {
    for (int f0 = lane; f0 < F; f0 += 64) {
      float gk = 0.f, gv = 0.f;
      const float ksv = MODE == EA_GATV2 ? ks[(s * H + h) * F + f0] : 0.f;
      for (int pt = b; pt < e; ++pt) {
        const int p = perm_t[pt];
        const int64_t dst = col_t[pt];
        float a = alpha[(int64_t)p * H + h];
        if (thr) a = philox_word((unsigned long long)p * H + h, seed, stream_id) >= thr ? a * dscale : 0.f;
        gv += a * dout[(dst * H + h) * F + f0];
        const float g = de[(int64_t)p * H + h];
        if (MODE == EA_GATV2) {
          const float sv = ksv + qd[(dst * H + h) * F + f0];
          gk += g * att[h * F + f0] * (sv > 0.f ? 1.f : slope);
        } else {
          gk += g * qd[(dst * H + h) * F + f0] * scale;
        }
      }
      if (MODE == EA_GATV2) dks[(s * H + h) * F + f0] = gk + gv;
      else { dks[(s * H + h) * F + f0] = gk; dv[(s * H + h) * F + f0] = gv; }
    }
  }
}

// ------------------------------------------------------------------ FAConv (single head, scalar scores)
__global__ __launch_bounds__(256) void fa_fwd_kernel(const float* __restrict__ x, const float* __restrict__ x0,
                                                      const float* __restrict__ al, const float* __restrict__ ar,
                                                      const int* __restrict__ rowptr, const int* __restrict__ col,
                                                      const float* __restrict__ val, float* __restrict__ out,
                                                      float* __restrict__ coef, int64_t N, int F, float eps, unsigned thr,
                                                      float dscale, unsigned long long seed, unsigned long long stream_id) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const int b = rowptr[i], e = rowptr[i + 1];
  const float ari = ar[i];
  for (int p = b + lane; p < e; p += 64) coef[p] = tanhf(al[col[p]] + ari);      // pre-dropout tanh score
  for (int f0 = lane; f0 < F; f0 += 64) {
    float acc = 0.f;
    for (int p = b; p < e; ++p) {
      float a = tanhf(al[col[p]] + ari);
      if (thr) a = philox_word((unsigned long long)p, seed, stream_id) >= thr ? a * dscale : 0.f;
      acc += a * val[p] * x[(int64_t)col[p] * F + f0];
    }
    out[i * F + f0] = acc + eps * x0[i * F + f0];
  }
}

__global__ __launch_bounds__(256) void fa_bwd_dst_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                          const float* __restrict__ coef, const int* __restrict__ rowptr,
                                                          const int* __restrict__ col, const float* __restrict__ val,
                                                          float* __restrict__ de, float* __restrict__ dar, int64_t N, int F,
                                                          unsigned thr, float dscale, unsigned long long seed,
                                                          unsigned long long stream_id) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const int b = rowptr[i], e = rowptr[i + 1];
  const float* drow = dout + i * F;
  float sum = 0.f;
  for (int p = b; p < e; ++p) {
    const float* xr = x + (int64_t)col[p] * F;
    float d = 0.f;
    for (int f = lane; f < F; f += 64) d += drow[f] * xr[f];
    d = wave_sum(d) * val[p];
    if (thr) d = philox_word((unsigned long long)p, seed, stream_id) >= thr ? d * dscale : 0.f;
    const float t = coef[p];
    const float g = d * (1.f - t * t);                             // through tanh
    if (lane == 0) de[p] = g;
    sum += g;
  }
  if (lane == 0) dar[i] = sum;
}

__global__ __launch_bounds__(256) void fa_bwd_src_kernel(const float* __restrict__ dout, const float* __restrict__ coef,
                                                          const float* __restrict__ de, const float* __restrict__ dar,
                                                          const int* __restrict__ rowptr_t, const int* __restrict__ col_t,
                                                          const float* __restrict__ val_t, const int* __restrict__ perm_t,
                                                          const float* __restrict__ att_l, const float* __restrict__ att_r,
                                                          float* __restrict__ dx, float* __restrict__ dal, int64_t N, int F,
                                                          unsigned thr, float dscale, unsigned long long seed,
                                                          unsigned long long stream_id) {
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= N) return;
  const int b = rowptr_t[s], e = rowptr_t[s + 1];
  float g = 0.f;
  for (int pt = b + lane; pt < e; pt += 64) g += de[perm_t[pt]];
  g = wave_sum(g);
  if (lane == 0) dal[s] = g;
  const float gr = dar[s];
  for (int f0 = lane; f0 < F; f0 += 64) {
    float acc = g * att_l[f0] + gr * att_r[f0];                    // al = <x, att_l>, ar = <x, att_r>
    for (int pt = b; pt < e; ++pt) {
      const int p = perm_t[pt];
      float a = coef[p];
      if (thr) a = philox_word((unsigned long long)p, seed, stream_id) >= thr ? a * dscale : 0.f;
      acc += a * val_t[pt] * dout[(int64_t)col_t[pt] * F + f0];
    }
    dx[s * F + f0] = acc;
  }
}

template <int MODE>
int launch_fwd(const float* ks, const float* qd, const float* v, const float* att, const int32_t* rowptr, const int32_t* col,
               const float* bias, float* out, float* alpha, int64_t N, int H, int F, float slope, float scale, uint32_t thr,
               float dscale, uint64_t seed, uint64_t stream_id, hipStream_t s) {
  hipLaunchKernelGGL((edge_attn_fwd_kernel<MODE>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, ks, qd, v, att, rowptr, col,
                     bias, out, alpha, N, H, F, slope, scale, thr, dscale, (unsigned long long)seed,
                     (unsigned long long)stream_id);
  return isic_launch_status();
}

template <int MODE>
int launch_bwd(const float* dout, const float* ks, const float* qd, const float* v, const float* att, const float* alpha,
               const int32_t* rowptr, const int32_t* col, const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t,
               float* de, float* dqd, float* dks, float* dv, float* datt, int64_t N, int H, int F, float slope, float scale,
               uint32_t thr, float dscale, uint64_t seed, uint64_t stream_id, hipStream_t s) {
  int64_t blocks = (N + 3) / 4;
  if (blocks > 2048) blocks = 2048;                                // persistent: d att lives in registers across rows
  hipLaunchKernelGGL((edge_attn_bwd_dst_kernel<MODE>), dim3((unsigned)blocks), dim3(256), 0, s, dout, ks, qd, v, att, alpha,
                     rowptr, col, de, dqd, datt, N, H, F, slope, scale, thr, dscale, (unsigned long long)seed,
                     (unsigned long long)stream_id);
  hipLaunchKernelGGL((edge_attn_bwd_src_kernel<MODE>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, dout, ks, qd, att, alpha,
                     de, rowptr_t, col_t, perm_t, dks, dv, N, H, F, slope, scale, thr, dscale, (unsigned long long)seed,
                     (unsigned long long)stream_id);
  return isic_launch_status();
}

}  // namespace

extern "C" {

int isic_edge_attn_fwd(int mode, const float* ks, const float* qd, const float* v, const float* att, const int32_t* rowptr,
                       const int32_t* col, const float* bias, float* out, float* alpha, int64_t N, int H, int F,
                       float negative_slope, float scale, uint32_t drop_threshold, float drop_scale, uint64_t seed,
                       uint64_t stream_id, void* stream) {
  ISIC_CHECK_ARG(N >= 0 && H > 0 && F > 0 && (mode == EA_GATV2 || mode == EA_DOT));
  if (N == 0) return ISIC_OK;
  ISIC_CHECK_ARG(ks && qd && v && rowptr && col && out && alpha && (mode != EA_GATV2 || att));
  hipStream_t s = as_stream(stream);
  return mode == EA_GATV2 ? launch_fwd<EA_GATV2>(ks, qd, v, att, rowptr, col, bias, out, alpha, N, H, F, negative_slope, scale,
                                                 drop_threshold, drop_scale, seed, stream_id, s)
                          : launch_fwd<EA_DOT>(ks, qd, v, att, rowptr, col, bias, out, alpha, N, H, F, negative_slope, scale,
                                               drop_threshold, drop_scale, seed, stream_id, s);
}

int isic_edge_attn_bwd(int mode, const float* dout, const float* ks, const float* qd, const float* v, const float* att,
                       const float* alpha, const int32_t* rowptr, const int32_t* col, const int32_t* rowptr_t,
                       const int32_t* col_t, const int32_t* perm_t, float* de, float* dqd, float* dks, float* dv, float* datt,
                       int64_t N, int H, int F, float negative_slope, float scale, uint32_t drop_threshold, float drop_scale,
                       uint64_t seed, uint64_t stream_id, void* stream) {
  ISIC_CHECK_ARG(N >= 0 && H > 0 && F > 0 && (mode == EA_GATV2 || mode == EA_DOT));
  if (N == 0) return ISIC_OK;
  ISIC_CHECK_ARG(dout && ks && qd && v && alpha && rowptr && col && rowptr_t && col_t && perm_t && de && dqd && dks);
  ISIC_CHECK_ARG(mode == EA_GATV2 ? (att && datt) : (dv != nullptr));
  hipStream_t s = as_stream(stream);
  return mode == EA_GATV2 ? launch_bwd<EA_GATV2>(dout, ks, qd, v, att, alpha, rowptr, col, rowptr_t, col_t, perm_t, de, dqd,
                                                 dks, dv, datt, N, H, F, negative_slope, scale, drop_threshold, drop_scale,
                                                 seed, stream_id, s)
                          : launch_bwd<EA_DOT>(dout, ks, qd, v, att, alpha, rowptr, col, rowptr_t, col_t, perm_t, de, dqd, dks,
                                               dv, datt, N, H, F, negative_slope, scale, drop_threshold, drop_scale, seed,
                                               stream_id, s);
}

int isic_fa_fwd(const float* x, const float* x0, const float* al, const float* ar, const int32_t* rowptr, const int32_t* col,
                const float* val, float* out, float* coef, int64_t N, int F, float eps, uint32_t drop_threshold,
                float drop_scale, uint64_t seed, uint64_t stream_id, void* stream) {
  ISIC_CHECK_ARG(N >= 0 && F > 0);
  if (N == 0) return ISIC_OK;
  ISIC_CHECK_ARG(x && x0 && al && ar && rowptr && col && val && out && coef);
  hipLaunchKernelGGL(fa_fwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, as_stream(stream), x, x0, al, ar, rowptr, col,
                     val, out, coef, N, F, eps, drop_threshold, drop_scale, (unsigned long long)seed,
                     (unsigned long long)stream_id);
  return isic_launch_status();
}

int isic_fa_bwd(const float* dout, const float* x, const float* coef, const float* att_l, const float* att_r,
                const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* rowptr_t, const int32_t* col_t,
                const float* val_t, const int32_t* perm_t, float* de, float* dar, float* dal, float* dx, int64_t N, int F,
                uint32_t drop_threshold, float drop_scale, uint64_t seed, uint64_t stream_id, void* stream) {
  ISIC_CHECK_ARG(N >= 0 && F > 0);
  if (N == 0) return ISIC_OK;
  ISIC_CHECK_ARG(dout && x && coef && att_l && att_r && rowptr && col && val && rowptr_t && col_t && val_t && perm_t && de &&
                 dar && dal && dx);
  const dim3 grid((unsigned)((N + 3) / 4));
  hipLaunchKernelGGL(fa_bwd_dst_kernel, grid, dim3(256), 0, as_stream(stream), dout, x, coef, rowptr, col, val, de, dar, N, F,
                     drop_threshold, drop_scale, (unsigned long long)seed, (unsigned long long)stream_id);
  hipLaunchKernelGGL(fa_bwd_src_kernel, grid, dim3(256), 0, as_stream(stream), dout, coef, de, dar, rowptr_t, col_t, val_t,
                     perm_t, att_l, att_r, dx, dal, N, F, drop_threshold, drop_scale, (unsigned long long)seed,
                     (unsigned long long)stream_id);
  return isic_launch_status();
}

}  // extern "C"
