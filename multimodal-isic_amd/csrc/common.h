// Shared device/host helpers for libisic_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/isic_hip.h"

#define ISIC_WAVE 64

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define ISIC_CHECK_ARG(cond) \
  do {                       \
    if (!(cond)) return ISIC_ERR_BAD_ARG; \
  } while (0)

static inline int isic_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ISIC_OK : ISIC_ERR_LAUNCH;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// tanh for the GEMM epilogues (25.7 M evaluations per GraphMIL step in the attention heads' first Linear: ocml's tanhf is
// a third of that kernel's epilogue).  |x| < 0.25: odd Taylor polynomial to x^9 (truncation < 1e-8 relative); otherwise
// 1 - 2 / (exp(2|x|) + 1) on v_exp_f32: absolute error < 1.5e-7 everywhere, exact saturation to +-1.
__device__ __forceinline__ float isic_tanhf(float x) {
  const float ax = fabsf(x);
  if (ax < 0.25f) {
    const float x2 = x * x;
    return x * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
  }
  const float e = __expf(2.f * ax);                        // inf for large |x|: 2 / inf = 0
  return copysignf(1.f - 2.f / (e + 1.f), x);
}

// ---------------------------------------------------------------- per-device one-time host state
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count belong to a DEVICE, not to the process: a process
// that drives two GPUs must set the attribute on both.  State is therefore keyed by hipGetDevice() (the device the
// caller's stream belongs to: torch makes it current before it hands out a stream).
constexpr int ISIC_MAX_DEVICES = 16;
static inline int isic_current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return d;
}
struct IsicPerDeviceOnce {
  std::once_flag flag[ISIC_MAX_DEVICES];
  hipError_t rc[ISIC_MAX_DEVICES];
};
template <class Fn>
static inline hipError_t isic_once_per_device(IsicPerDeviceOnce& o, Fn fn) {
  const int d = isic_current_device() % ISIC_MAX_DEVICES;
  std::call_once(o.flag[d], [&] { o.rc[d] = fn(); });
  return o.rc[d];
}
static inline int isic_cu_count() {
  static int cus[ISIC_MAX_DEVICES];                        // 0 = not asked yet; the race is benign (same value)
  const int dev = isic_current_device();
  int n = cus[dev % ISIC_MAX_DEVICES];
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev % ISIC_MAX_DEVICES] = n;
  }
  return n;
}

__host__ __device__ static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserved)
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits for every
// global store the wave has in flight -- a full memory round trip when it follows an epilogue's output stores.
// Full 128-byte lines per store instruction for the register-only tile epilogues (round 3, tests/probes/probe_rw.hip: a
// nontemporal 16-byte store instruction whose lanes cover 64 bytes of a row streams at 3.2 TB/s, one that covers whole 128-byte
// lines at 5.4).  After the MFMAs lane (fg, fr) of a wave owns, of row fr, the 16 bytes at column group 8 fg of BOTH 64-byte
// halves t = 0, 1 of a 128-byte line, so an instruction that stores one half writes 16 rows x 64 B.  Adjacent lanes (rows fr,
// fr ^ 1; same fg) swap one vector instead: the even lane keeps its t = 0 part and receives the odd row's t = 0 part, the odd
// lane keeps t = 1 and receives the even row's t = 1.  Then
//   store A: data `a` at (row fr & ~1, half fr & 1)      store B: data `b` at (row fr | 1, half fr & 1)
// and each instruction writes 8 rows x 128 B.  Four v_mov_dpp (quad_perm 1,0,3,2) and twelve selects per pair of vectors.
__device__ __forceinline__ void isic_pair_rows(const u32x4& v0, const u32x4& v1, bool odd, u32x4& a, u32x4& b) {
  u32x4 recv;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned send = odd ? v0[e] : v1[e];
    recv[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, false);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = odd ? recv[e] : v0[e];
    b[e] = odd ? v1[e] : recv[e];
  }
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------- Philox4x32-10
// Bit-for-bit the generator restated in oracle/philox.py.
struct Philox4 {
  unsigned int x, y, z, w;
};
__host__ __device__ static inline Philox4 philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2,
                                                        unsigned int c3, unsigned int k0, unsigned int k1) {
  const unsigned int M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned long long p0 = (unsigned long long)M0 * c0;
    unsigned long long p1 = (unsigned long long)M1 * c2;
    unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0;
    unsigned int hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
    unsigned int n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o = {c0, c1, c2, c3};
  return o;
}
// word (i & 3) of block (i >> 2) of the (seed, stream) sequence
__host__ __device__ static inline unsigned int philox_word(unsigned long long i, unsigned long long seed,
                                                           unsigned long long stream) {
  unsigned long long blk = i >> 2;
  Philox4 r = philox4x32_10((unsigned int)blk, (unsigned int)(blk >> 32), (unsigned int)stream,
                            (unsigned int)(stream >> 32), (unsigned int)seed, (unsigned int)(seed >> 32));
  switch (i & 3) {
    case 0: return r.x;
    case 1: return r.y;
    case 2: return r.z;
    default: return r.w;
  }
}
// keep-mask for 4 consecutive elements starting at a multiple of 4
__device__ __forceinline__ Philox4 philox_block(unsigned long long blk, unsigned long long seed,
                                                unsigned long long stream) {
  return philox4x32_10((unsigned int)blk, (unsigned int)(blk >> 32), (unsigned int)stream,
                       (unsigned int)(stream >> 32), (unsigned int)seed, (unsigned int)(seed >> 32));
}
__host__ __device__ static inline unsigned int dropout_threshold(float p) {
  double t = floor((double)p * 4294967296.0);
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned int)t;
}
