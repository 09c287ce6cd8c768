// Probe: what does v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 sustain with nothing else in the way?  Every fp32 GEMM of the
// graph path (persistent LDS-DMA, row-panel, register-fed A^T B) levels off at 0.5-0.6 of the 157.3 TFLOP/s the guide gives
// for the exact-fp32 matrix path; is that the kernels or the instruction?
//   hipcc --offload-arch=gfx950 -O3 tests/probes/probe_mfma_f32.hip -o tests/probes/build/probe_mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  const int iters = 20000;
  for (int bpc = 1; bpc <= 2; ++bpc) {                     // 4 or 8 waves per CU (1 or 2 per SIMD)
    const int grid = cus * bpc;
    float t = timeit([&] { hipLaunchKernelGGL(k16<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    double fl = (double)grid * 4 * iters * 16 * 2048.0;
    printf("16x16x4 f32, 16 accumulators, %d waves/SIMD: %7.2f ms  %6.1f TFLOP/s\n", bpc, t, fl / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    fl = (double)grid * 4 * iters * 4 * 2048.0;
    printf("16x16x4 f32,  4 accumulators, %d waves/SIMD: %7.2f ms  %6.1f TFLOP/s\n", bpc, t, fl / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    fl = (double)grid * 4 * iters * 4 * 4096.0;
    printf("32x32x2 f32,  4 accumulators, %d waves/SIMD: %7.2f ms  %6.1f TFLOP/s\n", bpc, t, fl / t / 1e9);
  }
  return 0;
}
