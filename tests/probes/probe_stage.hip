// Probe: how fast can a CU bring bytes from L2 / HBM into LDS, by which path?  Every staging-heavy kernel of this build
// (conv_pgemm, gemm_f16, wgrad_c128, ...) sits at ~11-15 B/clk per CU of LDS-DMA staging when all 256 CUs stream
// (DESIGN.md section 4).  Is that the memory system or the LDS-DMA path?
//   D : global_load_lds_dwordx4 (LDS-DMA, 1 KB per wave instruction, K in flight per wave)
//   R : global_load_dwordx4 into registers, then ds_write_b128 (K in flight per wave)
//   H : half of the waves D, half R (do the two paths add up?)
// One persistent block per CU; a block re-reads its own REGION bytes `passes` times (REGION small: L2-resident;
// REGION = 16 MB: every pass comes from HBM / the Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 tests/probes/probe_stage.hip -o tests/probes/build/probe_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// MODE 0 = D, 1 = R, 2 = H.  WAVES waves per block, K chunks (1 KB per wave) in flight per wave.
template <int MODE, int WAVES, int K>
__global__ __launch_bounds__(WAVES * 64) void k_stage(const unsigned char* __restrict__ src, long region, int passes,
                                                      unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned char* base = src + (long)blockIdx.x * region;
  const long chunks = region / 1024;                       // 1 KB chunks of the block's region
  const bool dma = MODE == 0 || (MODE == 2 && (wave & 1) == 0);
  unsigned acc = 0;
  for (int p = 0; p < passes; ++p) {
    for (long c0 = wave; c0 < chunks; c0 += (long)WAVES * K) {
      if (dma) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const long c = c0 + (long)k * WAVES;
          const unsigned char* g = base + (c < chunks ? c : 0) * 1024 + lane * 16;
          glds16(g, lds0 + (unsigned)((wave * K + k) * 1024));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        u32x4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const long c = c0 + (long)k * WAVES;
          v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (c < chunks ? c : 0) * 1024 + lane * 16));
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {      // inline asm: the compiler removed these "dead" LDS stores -- and the loads with them
          const unsigned addr = lds0 + (unsigned)((wave * K + k) * 1024 + lane * 16);
          asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v[k]) : "memory");
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  acc = *reinterpret_cast<unsigned*>(smem + ((wave * K) * 1024 + lane * 16));
  if (acc == 0x12345678u) sink[0] = acc;                   // keeps the LDS traffic alive
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 3; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}

template <int MODE, int WAVES, int K>
int run(const unsigned char* src, long region, int passes, unsigned* sink, int cus, const char* what) {
  const int lds = WAVES * K * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stage<MODE, WAVES, K>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  float t = timeit([&] { hipLaunchKernelGGL((k_stage<MODE, WAVES, K>), dim3(cus), dim3(WAVES * 64), lds, 0, src, region, passes, sink); });
  const double bytes = (double)cus * region * passes;
  printf("%-10s mode %c waves %2d in-flight %2d KB/wave : %8.3f ms  %6.2f TB/s  %5.1f B/clk/CU (at 2.4 GHz)\n", what,
         "DRH"[MODE], WAVES, K, t, bytes / t / 1e9, bytes / (t * 1e-3) / cus / 2.4e9);
  return 0;
}

int main() {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const long big = 16L << 20;                               // 16 MB per block: 4 GB in all, streamed from HBM
  unsigned char* src; unsigned* sink;
  CK(hipMalloc(&src, (long)cus * big)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(src, 0x3c, (long)cus * big));
  printf("%d CUs\n", cus);
  // ---- L2-resident: 64 KB per block (16 MB in all), 256 passes
#define L2RUN(M, W, K) if (run<M, W, K>(src, 64L << 10, 256, sink, cus, "L2 64KB")) return 1;
#define HBMRUN(M, W, K) if (run<M, W, K>(src, big, 1, sink, cus, "HBM 16MB")) return 1;
  L2RUN(0, 4, 4) L2RUN(0, 8, 4) L2RUN(0, 8, 8) L2RUN(0, 16, 4)
  L2RUN(1, 4, 4) L2RUN(1, 8, 4) L2RUN(1, 8, 8) L2RUN(1, 16, 4)
  L2RUN(2, 8, 4) L2RUN(2, 16, 4) L2RUN(2, 16, 8)
  HBMRUN(0, 8, 4) HBMRUN(0, 16, 4) HBMRUN(1, 8, 4) HBMRUN(1, 16, 4) HBMRUN(1, 16, 8) HBMRUN(2, 16, 4) HBMRUN(2, 16, 8)
  return 0;
}
