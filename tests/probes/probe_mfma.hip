// Standalone hardware probe (test infrastructure): verifies, with exact integer data, the MFMA operand /
// accumulator lane maps and the ds_read_b64_tr_b16 gather semantics that the conv kernels rely on.
//   hipcc --offload-arch=gfx950 -O2 tests/probes/probe_mfma.hip -o tests/probes/build/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

// C[16x16] = A[16x32] * B[32x16], A row-major [m][k], Bt row-major [n][k]
__global__ void k16(const float* A, const float* Bt, float* C) {
  int l = threadIdx.x;
  s16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (short)f2bf(A[(l & 15) * 32 + 8 * (l >> 4) + j]);
    b[j] = (short)f2bf(Bt[(l & 15) * 32 + 8 * (l >> 4) + j]);
  }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
// C[32x32] = A[32x16] * B[16x32]
__global__ void k32(const float* A, const float* Bt, float* C) {
  int l = threadIdx.x;
  s16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (short)f2bf(A[(l & 31) * 16 + 8 * (l >> 5) + j]);
    b[j] = (short)f2bf(Bt[(l & 31) * 16 + 8 * (l >> 5) + j]);
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
// f32 16x16x4: C = A[16x4] * B[4x16]
__global__ void kf32(const float* A, const float* Bt, float* C) {
  int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], Bt[(l & 15) * 4 + (l >> 4)], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
// LDS tile [16 rows(k)][64 cols(m)] of u16 = row*64+col.  Group g (16 lanes) reads block rows 4g..4g+3?
// Here: every lane 4q+p of a group supplies &tile[r0+q][c0+4p]; report what lane gets.
__global__ void ktr(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[16 * 64];
  int l = threadIdx.x;
  for (int i = l; i < 16 * 64; i += 64) tile[i] = (unsigned short)i;
  __syncthreads();
  int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  int r0 = 4 * g, c0 = 16;  // group g reads rows 4g..4g+3, cols 16..31
  const unsigned short* addr = &tile[(r0 + q) * 64 + c0 + 4 * p];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

int main() {
  int fails = 0;
  {
    std::vector<float> A(16 * 32), Bt(16 * 32), C(256), R(256, 0.f);
    for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7) % 11 - 5); Bt[i] = (float)((i * 5) % 13 - 6); }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) for (int k = 0; k < 32; ++k) R[m * 16 + n] += A[m * 32 + k] * Bt[n * 32 + k];
    float *dA, *dB, *dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 2048, hipMemcpyHostToDevice);
    k16<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += C[i] != R[i];
    printf("mfma_16x16x32_bf16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  }
  {
    std::vector<float> A(32 * 16), Bt(32 * 16), C(1024), R(1024, 0.f);
    for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7) % 11 - 5); Bt[i] = (float)((i * 5) % 13 - 6); }
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) R[m * 32 + n] += A[m * 16 + k] * Bt[n * 16 + k];
    float *dA, *dB, *dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 2048, hipMemcpyHostToDevice);
    k32<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += C[i] != R[i];
    printf("mfma_32x32x16_bf16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  }
  {
    std::vector<float> A(64), Bt(64), C(256), R(256, 0.f);
    for (int i = 0; i < 64; ++i) { A[i] = (float)((i * 7) % 11 - 5); Bt[i] = (float)((i * 5) % 13 - 6); }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) for (int k = 0; k < 4; ++k) R[m * 16 + n] += A[m * 4 + k] * Bt[n * 4 + k];
    float *dA, *dB, *dC;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 256, hipMemcpyHostToDevice);
    kf32<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += C[i] != R[i];
    printf("mfma_16x16x4_f32 layout: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  }
  {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    std::vector<unsigned short> o(256);
    ktr<<<1, 64>>>(d);
    hipMemcpy(o.data(), d, 512, hipMemcpyDeviceToHost);
    // expectation: lane (g,i) element q == tile[4g+q][16+i] == (4g+q)*64 + 16 + i
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) bad += o[l * 4 + q] != (unsigned short)((4 * (l >> 4) + q) * 64 + 16 + (l & 15));
    printf("ds_read_tr16_b64 semantics: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
    if (bad) for (int l = 0; l < 20; ++l) printf("  lane %d: %d %d %d %d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
  }
  hipDeviceSynchronize();
  return fails ? 1 : 0;
}
