// Probe: pure-read and pure-write streaming rates of the chip (what bounds the load / store PHASES of the tile kernels:
// DESIGN.md section 6 prices them at 6.4 / 3.2 TB/s from read+write passes).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tests/probes/probe_rw.hip -o /tmp/probe_rw && /tmp/probe_rw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT, int VPT>
__global__ __launch_bounds__(256) void k_write(u32x4* __restrict__ y, long n, unsigned v) {
  const long base = ((long)blockIdx.x * 256 * VPT) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const long j = base + (long)u * 256;
    if (j < n) { const u32x4 o = {v, v + 1, v + 2, v + 3}; if (NT) __builtin_nontemporal_store(o, y + j); else y[j] = o; }
  }
}
template <int NT, int VPT>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ x, unsigned* __restrict__ sink, long n) {
  const long base = ((long)blockIdx.x * 256 * VPT) + threadIdx.x;
  unsigned acc = 0;
#pragma unroll
  for (int u = 0; u < VPT; ++u) {
    const long j = base + (long)u * 256;
    if (j < n) { const u32x4 v = NT ? __builtin_nontemporal_load(x + j) : x[j]; acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
  }
  if (acc == 0x12345678u) *sink = acc;
}
// a "tile epilogue"-shaped write: each wave writes 16 rows x 256 B segments (row pitch 768 B): what a 128-column fp16 slice does
template <int NT>
__global__ __launch_bounds__(256) void k_write_rows(u32x4* __restrict__ y, long rows, int pitch_vec, unsigned v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long r0 = ((long)blockIdx.x * 4 + wave) * 16; r0 < rows; r0 += (long)gridDim.x * 64) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long r = r0 + i * 4 + (lane >> 4);
      const u32x4 o = {v, v + 1, v + 2, v + 3};
      if (r < rows) { if (NT) __builtin_nontemporal_store(o, y + r * pitch_vec + (lane & 15)); else y[r * pitch_vec + (lane & 15)] = o; }
    }
  }
}

// SEG-byte segments per row and store instruction (SEG / 16 lanes per row, 1024 / SEG rows per instruction), the segments of one
// row written by CONSECUTIVE instructions of the same wave until `width` bytes of the row are covered -- the shape of a tile
// epilogue whose lane owns 16 bytes of a row: SEG = 64 is gemm_f16 / conv_halo today (4 lanes of a row per instruction)
template <int SEG, int NT>
__global__ __launch_bounds__(256) void k_write_seg(unsigned char* __restrict__ y, long rows, int pitch, int width, unsigned v) {
  constexpr int LPR = SEG / 16, RPI = 64 / LPR;                 // lanes per row, rows per instruction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane / LPR, lc = lane % LPR;
  for (long r0 = ((long)blockIdx.x * 4 + wave) * RPI; r0 < rows; r0 += (long)gridDim.x * 4 * RPI) {
    const long r = r0 + lr;
    for (int c = 0; c < width; c += SEG) {
      const u32x4 o = {v, v + 1, v + 2, v + 3};
      u32x4* dst = reinterpret_cast<u32x4*>(y + r * pitch + c + lc * 16);
      if (r < rows) { if (NT) __builtin_nontemporal_store(o, dst); else *dst = o; }
    }
  }
}

int main() {
  const long bytes = 2L << 30, n = bytes / 16;
  u32x4* buf; unsigned* sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
  hipMemset(buf, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch, const char* what) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-56s %.3f ms  %.2f TB/s\n", what, ms, bytes / ms / 1e9);
  };
#define W(NT, VPT) time([&] { hipLaunchKernelGGL((k_write<NT, VPT>), dim3((n + 256 * VPT - 1) / (256 * VPT)), dim3(256), 0, 0, buf, n, 7u); }, "write  nt=" #NT " vec/thread=" #VPT)
#define R(NT, VPT) time([&] { hipLaunchKernelGGL((k_read<NT, VPT>), dim3((n + 256 * VPT - 1) / (256 * VPT)), dim3(256), 0, 0, buf, sink, n); }, "read   nt=" #NT " vec/thread=" #VPT)
  W(0, 1); W(1, 1); W(0, 4); W(1, 4); W(1, 8);
  R(0, 1); R(1, 1); R(0, 4); R(1, 4); R(1, 8);
  time([&] { hipMemsetAsync(buf, 0, bytes, 0); }, "hipMemsetAsync");
  const long rows = bytes / 768;       // rows of 384 fp16, a wave writes 256-byte segments of 16 rows
  time([&] { hipLaunchKernelGGL((k_write_rows<1>), dim3(2048), dim3(256), 0, 0, buf, rows, 48, 7u); }, "write 256-B row segments (1/3 of the rows' bytes) nt=1 (x3)");
  time([&] { hipLaunchKernelGGL((k_write_rows<0>), dim3(2048), dim3(256), 0, 0, buf, rows, 48, 7u); }, "write 256-B row segments (1/3 of the rows' bytes) nt=0 (x3)");
  {
    const long rows2 = bytes / 768;
    unsigned char* yb = reinterpret_cast<unsigned char*>(buf);
#define S(SEG, NT, WIDTH) time([&] { hipLaunchKernelGGL((k_write_seg<SEG, NT>), dim3(4096), dim3(256), 0, 0, yb, rows2, 768, WIDTH, 7u); }, \
                               "rows of 768 B: " #WIDTH " B of each in " #SEG "-B segments nt=" #NT " (TB/s x " #WIDTH "/768)")
    S(32, 1, 256); S(64, 1, 256); S(128, 1, 256); S(256, 1, 256);
    S(32, 0, 256); S(64, 0, 256); S(128, 0, 256); S(256, 0, 256);
    S(64, 1, 768); S(128, 1, 768); S(256, 1, 768);
  }
  return 0;
}
