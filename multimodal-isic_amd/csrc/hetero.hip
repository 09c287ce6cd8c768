// Edge-wise heterophily measures of a patch graph (gfx950): per directed edge (src -> dst)
//   H_kl        = sum_c p[src,c] * log((p[src,c] + eps) / (p[dst,c] + eps))        (04_measure_heterophily.py:164)
//   H_dirichlet = 0.5 * || x[src] - x[dst] ||^2                                     (:165)
//   H_spatial   = Euclidean distance of the two patches on the grid_w-wide lattice  (:124-125,166)
//   same_class  = dominant_class[src] == dominant_class[dst]                        (:130)
// HBM-bound row gather: one wave per edge, both embedding rows read with 16-byte loads (D floats each), squared
// difference reduced across the 64 lanes; the class-probability rows (C <= 64) ride on the first lanes.
// Self loops are NOT dropped here (the reference strips them, :117-118): the host filters on src != dst so that
// the surviving values keep the reference's edge order.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void edge_hetero_kernel(const float* __restrict__ x, const float* __restrict__ probs,
                                                          const int* __restrict__ dominant, const int64_t* __restrict__ src,
                                                          const int64_t* __restrict__ dst, int64_t E, int D, int C, int grid_w,
                                                          int nodes_per_graph, float eps, float* __restrict__ h_kl,
                                                          float* __restrict__ h_dir, float* __restrict__ h_spatial,
                                                          float* __restrict__ same_class) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t e = wave0; e < E; e += nwaves) {
    const int64_t s = src[e], d = dst[e];
    const float* xs = x + (size_t)s * D;
    const float* xd = x + (size_t)d * D;
    float acc = 0.f;
    if ((D & 3) == 0) {
      for (int i = lane * 4; i < D; i += 256) {
        const float4 a = *reinterpret_cast<const float4*>(xs + i);
        const float4 b = *reinterpret_cast<const float4*>(xd + i);
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        acc += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    } else {
      for (int i = lane; i < D; i += 64) {
        const float df = xs[i] - xd[i];
        acc += df * df;
      }
    }
    float kl = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float ps = probs[(size_t)s * C + c], pd = probs[(size_t)d * C + c];
      kl += ps * logf((ps + eps) / (pd + eps));
    }
    acc = wave_sum(acc);
    kl = wave_sum(kl);
    if (lane == 0) {
      h_dir[e] = 0.5f * acc;
      h_kl[e] = kl;
      // lattice coordinates of the node inside ITS graph (04:124-125 uses % 14 and // 14 of the local index)
      const int ls = (int)(s % nodes_per_graph), ld = (int)(d % nodes_per_graph);
      const float dx = (float)(ls % grid_w - ld % grid_w), dy = (float)(ls / grid_w - ld / grid_w);
      h_spatial[e] = sqrtf(dx * dx + dy * dy);
      same_class[e] = dominant[s] == dominant[d] ? 1.f : 0.f;
    }
  }
}

}  // namespace

extern "C" {

int isic_edge_heterophily_f32(const float* x, const float* probs, const int32_t* dominant_class, const int64_t* src,
                              const int64_t* dst, int64_t num_edges, int D, int C, int grid_w, int nodes_per_graph,
                              float eps, float* h_kl, float* h_dirichlet, float* h_spatial, float* same_class,
                              void* stream) {
  ISIC_CHECK_ARG(num_edges >= 0 && D > 0 && C > 0 && grid_w > 0 && nodes_per_graph > 0);
  if (num_edges == 0) return ISIC_OK;
  ISIC_CHECK_ARG(x && probs && dominant_class && src && dst && h_kl && h_dirichlet && h_spatial && same_class);
  int64_t blocks = (num_edges + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(edge_hetero_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, probs, dominant_class,
                     src, dst, num_edges, D, C, grid_w, nodes_per_graph, eps, h_kl, h_dirichlet, h_spatial, same_class);
  return isic_launch_status();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Lesion-mask -> patch flags (save_latent.py:73-87): flags[b, i, j] = 1 iff the P x P pixel block (i, j) of mask b
// holds any value > 0.  One wave per patch; lanes stride the block's pixels; a ballot reduces.
namespace {

__global__ __launch_bounds__(256) void mask_patch_flags_kernel(const float* __restrict__ mask, unsigned char* __restrict__ flags,
                                                               int64_t B, int H, int W, int P) {
  const int lane = threadIdx.x & 63;
  const int gh = H / P, gw = W / P;
  const int64_t total = B * gh * gw;
  for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < total; w += (int64_t)gridDim.x * 4) {
    const int64_t b = w / (gh * gw);
    const int r = (int)(w - b * (gh * gw));
    const int pi = r / gw, pj = r - pi * gw;
    const float* base = mask + (b * H + (int64_t)pi * P) * W + (int64_t)pj * P;
    bool any = false;
    for (int t = lane; t < P * P; t += 64) any |= base[(int64_t)(t / P) * W + (t % P)] > 0.f;
    const unsigned long long bal = __ballot(any);
    if (lane == 0) flags[w] = bal != 0ull ? 1 : 0;
  }
}

}  // namespace

extern "C" int isic_mask_patch_flags_f32(const float* mask, uint8_t* flags, int64_t B, int H, int W, int patch, void* stream) {
  ISIC_CHECK_ARG(B >= 0 && H > 0 && W > 0 && patch > 0 && H % patch == 0 && W % patch == 0);
  if (B == 0) return ISIC_OK;
  ISIC_CHECK_ARG(mask && flags);
  int64_t blocks = (B * (H / patch) * (W / patch) + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mask_patch_flags_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), mask, flags, B, H, W,
                     patch);
  return isic_launch_status();
}
