// Probe: how fast can an elementwise bf16 pass (read 16 B, scale/shift/ReLU, write 16 B) stream on this chip, as a
// function of launch shape and cache hints?  822 MB in, 822 MB out (ResNet-18 layer1 activation at 2048 images).
//   hipcc --offload-arch=gfx950 -O3 tests/probes/probe_stream.hip -o tests/probes/build/probe_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ u32x4 work(u32x4 v, float sc, float sh) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = __uint_as_float(v[i] << 16), b = __uint_as_float(v[i] & 0xFFFF0000u);
    a = fmaxf(a * sc + sh, 0.f); b = fmaxf(b * sc + sh, 0.f);
    o[i] = (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xFFFF0000u);
  }
  return o;
}
// MODE bit0: nontemporal load, bit1: nontemporal store; UNR vectors per thread per iteration; grid-stride
template <int MODE, int UNR>
__global__ __launch_bounds__(256) void k_stride(const u32x4* __restrict__ x, u32x4* __restrict__ y, long n, float sc, float sh) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * UNR) {
    u32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long j = i + u * stride;
      if (j < n) v[u] = (MODE & 1) ? __builtin_nontemporal_load(x + j) : x[j];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long j = i + u * stride;
      if (j < n) { const u32x4 o = work(v[u], sc, sh); if (MODE & 2) __builtin_nontemporal_store(o, y + j); else y[j] = o; }
    }
  }
}
// one contiguous chunk of UNR vectors per thread (block-contiguous), no loop
template <int MODE, int UNR>
__global__ __launch_bounds__(256) void k_flat(const u32x4* __restrict__ x, u32x4* __restrict__ y, long n, float sc, float sh) {
  const long base = ((long)blockIdx.x * UNR) * 256 + threadIdx.x;
  u32x4 v[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) { const long j = base + u * 256; if (j < n) v[u] = (MODE & 1) ? __builtin_nontemporal_load(x + j) : x[j]; }
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const long j = base + u * 256;
    if (j < n) { const u32x4 o = work(v[u], sc, sh); if (MODE & 2) __builtin_nontemporal_store(o, y + j); else y[j] = o; }
  }
}
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  const long n = 2048L * 56 * 56 * 64 / 8;      // 16-byte vectors
  u32x4 *x, *y;
  CK(hipMalloc(&x, n * 16)); CK(hipMalloc(&y, n * 16));
  CK(hipMemset(x, 0x3c, n * 16));
  const double gb = 2.0 * n * 16 / 1e9;
#define RUN_S(M, U, G) { float t = timeit([&] { hipLaunchKernelGGL((k_stride<M, U>), dim3(G), dim3(256), 0, 0, x, y, n, 1.1f, 0.1f); }); \
    printf("grid-stride  nt(load,store)=(%d,%d) unroll %d grid %6d : %.3f ms  %.2f TB/s\n", M & 1, (M >> 1) & 1, U, G, t, gb / t); }
#define RUN_F(M, U) { const int G = (int)((n + 256L * U - 1) / (256L * U)); float t = timeit([&] { hipLaunchKernelGGL((k_flat<M, U>), dim3(G), dim3(256), 0, 0, x, y, n, 1.1f, 0.1f); }); \
    printf("flat         nt(load,store)=(%d,%d) %d vec/thread grid %6d : %.3f ms  %.2f TB/s\n", M & 1, (M >> 1) & 1, U, G, t, gb / t); }
  RUN_S(2, 2, 8192) RUN_S(3, 2, 8192) RUN_S(0, 2, 8192) RUN_S(1, 2, 8192)
  RUN_S(2, 4, 8192) RUN_S(3, 4, 8192) RUN_S(2, 4, 4096) RUN_S(2, 4, 2048) RUN_S(2, 8, 2048) RUN_S(3, 8, 2048) RUN_S(2, 2, 16384) RUN_S(2, 1, 32768)
  RUN_F(2, 1) RUN_F(2, 2) RUN_F(2, 4) RUN_F(3, 4) RUN_F(0, 4) RUN_F(2, 8) RUN_F(3, 8)
  return 0;
}
