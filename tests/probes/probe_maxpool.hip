// Probe: variants of the fused BatchNorm+ReLU+3x3/2 max-pool forward (the stem's largest HBM pass), timed back to back.
//   hipcc --offload-arch=gfx950 -O3 tests/probes/probe_maxpool.hip -o tests/probes/build/probe_maxpool
#include <cstdio>
#include <vector>
#include "../../multimodal-isic_amd/csrc/pool_grad.h"
using isic_pool::pack8;
using isic_pool::unpack8;

#define ARGS const unsigned short* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, \
    unsigned short* __restrict__ y, unsigned char* __restrict__ argmax, unsigned short* __restrict__ xsel, int N, int H, int W, int C, int Ho, int Wo

__device__ __forceinline__ void finish(int64_t i, float (&best)[8], float (&bx)[8], unsigned char (&bi)[8], unsigned short* y,
                                       unsigned char* argmax, unsigned short* xsel) {
  __builtin_nontemporal_store(pack8(best), reinterpret_cast<u32x4*>(y + i * 8));
  if (xsel) __builtin_nontemporal_store(pack8(bx), reinterpret_cast<u32x4*>(xsel + i * 8));
  if (argmax) {
    u32x2 p;
    p[0] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
    p[1] = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
    __builtin_nontemporal_store(p, reinterpret_cast<u32x2*>(argmax + i * 8));
  }
}

// V0: loads behind bounds branches (the round-1 kernel)
__global__ __launch_bounds__(256) void v0(ARGS) {
  const int cgs = C >> 3;
  const int64_t nvec = (int64_t)N * Ho * Wo * cgs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cgs);
    int64_t t = i / cgs;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float best[8], bx[8], sc[8], sh[8];
    unsigned char bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bx[j] = 0.f; bi[j] = 0; sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j]; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = ho * 2 - 1 + kh;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = wo * 2 - 1 + kw;
        if (wi < 0 || wi >= W) continue;
        float f[8], r[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + (((int64_t)n * H + hi) * W + wi) * C + cg * 8), r);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(r[j] * sc[j] + sh[j], 0.f);
        unpack8(pack8(f), f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (f[j] > best[j]) { best[j] = f[j]; bx[j] = r[j]; bi[j] = (unsigned char)(kh * 3 + kw); }
      }
    }
    finish(i, best, bx, bi, y, argmax, xsel);
  }
}

// V1/V2: nine clamped loads up front; V2 caps the registers at 128 (4 waves per SIMD)
template <int DUMMY>
__device__ __forceinline__ void body_clamped(ARGS) {
  const int cgs = C >> 3;
  const int64_t nvec = (int64_t)N * Ho * Wo * cgs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cgs);
    int64_t t = i / cgs;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    u32x4 raw[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = min(max(ho * 2 - 1 + kh, 0), H - 1);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = min(max(wo * 2 - 1 + kw, 0), W - 1);
        raw[kh * 3 + kw] = *reinterpret_cast<const u32x4*>(x + (((int64_t)n * H + hi) * W + wi) * C + cg * 8);
      }
    }
    float best[8], bx[8], sc[8], sh[8];
    unsigned char bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bx[j] = 0.f; bi[j] = 0; sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j]; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = ho * 2 - 1 + kh;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = wo * 2 - 1 + kw;
        const bool in = hi >= 0 && hi < H && wi >= 0 && wi < W;
        float f[8], r[8];
        unpack8(raw[kh * 3 + kw], r);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(r[j] * sc[j] + sh[j], 0.f);
        unpack8(pack8(f), f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (in && f[j] > best[j]) { best[j] = f[j]; bx[j] = r[j]; bi[j] = (unsigned char)(kh * 3 + kw); }
      }
    }
    finish(i, best, bx, bi, y, argmax, xsel);
  }
}
__global__ __launch_bounds__(256) void v1(ARGS) { body_clamped<0>(x, scale, shift, y, argmax, xsel, N, H, W, C, Ho, Wo); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void v2(ARGS) {
  body_clamped<1>(x, scale, shift, y, argmax, xsel, N, H, W, C, Ho, Wo);
}

// V3: block = 2 output rows x 56 columns of one image; the 5 input rows are normalised ONCE (each input element is
// loaded once, coalesced, all loads independent), kept in LDS as bf16 pairs (z, raw) and pooled from there
constexpr int V3_R = 2;
__global__ __launch_bounds__(512) void v3(ARGS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [(2R+1)][W][C] z then raw
  const int cgs = C >> 3, tid = threadIdx.x;
  const int strips = (Ho + V3_R - 1) / V3_R;
  const int n = blockIdx.x / strips, st = blockIdx.x % strips;
  const int ho0 = st * V3_R, hi0 = ho0 * 2 - 1, rows = 2 * V3_R + 1;
  const int rowv = W * cgs;                     // 16-byte vectors per input row
  float sc[8], sh[8];
  const int cg = tid % cgs;                     // 512 % cgs == 0: a thread keeps its channel group
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j]; }
  unsigned char* zs = lds;
  unsigned char* rs = lds + (size_t)rows * rowv * 16;
  for (int v = tid; v < rows * rowv; v += 512) {
    const int r = v / rowv, c = v - r * rowv;
    const int hi = hi0 + r;
    if (hi < 0 || hi >= H) continue;
    const u32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + ((int64_t)n * H + hi) * W * C) + c);
    float f[8];
    unpack8(raw, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j], 0.f);
    *reinterpret_cast<u32x4*>(zs + (size_t)v * 16) = pack8(f);
    *reinterpret_cast<u32x4*>(rs + (size_t)v * 16) = raw;
  }
  __syncthreads();
  for (int o = tid; o < V3_R * Wo * cgs; o += 512) {
    const int ocg = o % cgs, wo = (o / cgs) % Wo, hl = o / (cgs * Wo);
    const int ho = ho0 + hl;
    if (ho >= Ho) continue;
    float best[8], bx[8];
    unsigned char bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bx[j] = 0.f; bi[j] = 0; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = ho * 2 - 1 + kh, r = hi - hi0;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = wo * 2 - 1 + kw;
        const bool in = hi >= 0 && hi < H && wi >= 0 && wi < W;
        const int wic = min(max(wi, 0), W - 1);
        float f[8], rr[8];
        unpack8(*reinterpret_cast<const u32x4*>(zs + ((size_t)(r * W + wic) * cgs + ocg) * 16), f);
        unpack8(*reinterpret_cast<const u32x4*>(rs + ((size_t)(r * W + wic) * cgs + ocg) * 16), rr);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (in && f[j] > best[j]) { best[j] = f[j]; bx[j] = rr[j]; bi[j] = (unsigned char)(kh * 3 + kw); }
      }
    }
    const int64_t i = (((int64_t)n * Ho + ho) * Wo + wo) * cgs + ocg;
    finish(i, best, bx, bi, y, argmax, xsel);
  }
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 1024, H = 112, W = 112, C = 64, Ho = 56, Wo = 56;
  const size_t nx = (size_t)N * H * W * C, ny = (size_t)N * Ho * Wo * C;
  unsigned short *x, *y, *xs, *y_ref, *xs_ref; unsigned char *am, *am_ref; float *sc, *sh;
  CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&y, ny * 2)); CK(hipMalloc(&xs, ny * 2)); CK(hipMalloc(&am, ny));
  CK(hipMalloc(&y_ref, ny * 2)); CK(hipMalloc(&xs_ref, ny * 2)); CK(hipMalloc(&am_ref, ny));
  CK(hipMalloc(&sc, 256)); CK(hipMalloc(&sh, 256));
  {
    std::vector<unsigned short> hx(1 << 22);
    unsigned s = 12345;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3F00 + ((s >> 16) & 0xFF) + ((s >> 31) << 15)); }
    for (size_t o = 0; o < nx; o += hx.size()) CK(hipMemcpy(x + o, hx.data(), std::min(hx.size(), nx - o) * 2, hipMemcpyHostToDevice));
    float hs[64], hh[64];
    for (int c = 0; c < 64; ++c) { hs[c] = (c % 5 == 0 ? -1.f : 1.f) * (0.5f + c / 64.f); hh[c] = 0.1f * (c % 7) - 0.3f; }
    CK(hipMemcpy(sc, hs, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(sh, hh, 256, hipMemcpyHostToDevice));
  }
  const int64_t nvec = (int64_t)N * Ho * Wo * 8;
  auto run = [&](int v, unsigned short* yy, unsigned char* aa, unsigned short* ss) {
    const int g8 = (int)std::min<int64_t>((nvec + 255) / 256, 8192), g16 = (int)std::min<int64_t>((nvec + 255) / 256, 16384);
    if (v == 0) hipLaunchKernelGGL(v0, dim3(g8), dim3(256), 0, 0, x, sc, sh, yy, aa, ss, N, H, W, C, Ho, Wo);
    if (v == 1) hipLaunchKernelGGL(v1, dim3(g16), dim3(256), 0, 0, x, sc, sh, yy, aa, ss, N, H, W, C, Ho, Wo);
    if (v == 2) hipLaunchKernelGGL(v2, dim3(g16), dim3(256), 0, 0, x, sc, sh, yy, aa, ss, N, H, W, C, Ho, Wo);
    if (v == 3) {
      const int lds = 2 * (2 * V3_R + 1) * W * C * 2;
      hipFuncSetAttribute(reinterpret_cast<const void*>(v3), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(v3, dim3(N * ((Ho + V3_R - 1) / V3_R)), dim3(512), lds, 0, x, sc, sh, yy, aa, ss, N, H, W, C, Ho, Wo);
    }
  };
  run(0, y_ref, am_ref, xs_ref);
  CK(hipDeviceSynchronize());
  std::vector<unsigned short> a(ny), b(ny);
  std::vector<unsigned char> ca(ny), cb(ny);
  CK(hipMemcpy(a.data(), y_ref, ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ca.data(), am_ref, ny, hipMemcpyDeviceToHost));
  std::vector<unsigned short> sa(ny), sb(ny);
  CK(hipMemcpy(sa.data(), xs_ref, ny * 2, hipMemcpyDeviceToHost));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double gb = (nx * 2 + ny * 5) / 1e9;
  for (int v = 0; v < 4; ++v) {
    CK(hipMemset(y, 0, ny * 2)); CK(hipMemset(am, 0, ny)); CK(hipMemset(xs, 0, ny * 2));
    run(v, y, am, xs);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(b.data(), y, ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(cb.data(), am, ny, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sb.data(), xs, ny * 2, hipMemcpyDeviceToHost));
    const bool same = a == b && ca == cb && sa == sb;
    CK(hipEventRecord(e0));
    for (int it = 0; it < 5; ++it) run(v, y, am, xs);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("variant %d: %.3f ms  %.2f TB/s compulsory  identical to v0: %s\n", v, ms, gb / ms, same ? "yes" : "NO");
  }
  return 0;
}
