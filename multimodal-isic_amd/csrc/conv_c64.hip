// 3x3 / stride 1 / pad 1 convolution for the 64 -> 64 channel layers (ResNet-18 layer1, the largest
// activations of the network) with the INPUT PATCH RESIDENT IN LDS (gfx950, bf16 MFMA).
//
// The generic implicit GEMM re-fetches every input pixel once per tap: 9 x 128 B per output pixel
// from L2 for only 64 output channels -- 52 FLOP per staged byte, an L2-bandwidth-bound kernel.
// Here a block owns an 8 x 32 output tile of one image, stages the (8+2) x (32+2) pixel halo patch
// (43.5 KB) ONCE by LDS-DMA and reads all nine taps' A-fragments out of it with per-lane shifted
// ds_read_b128 addresses; only the 8 KB weight slice of the next tap streams in (two stages) while
// the current tap is multiplied: 3.1x fewer bytes staged per output pixel, 2 blocks per CU.
// Used for the forward pass and (flipped weights) the data gradient of those layers, with the same
// fused epilogues as conv_igemm.hip: residual-gradient addend, BatchNorm sum / sum-of-squares.
//
// Same operand conventions as conv_igemm.hip (NHWC bf16 activations, [64][3][3][64] bf16 weights,
// XOR-swizzled 128-byte LDS rows written lane-linearly by the DMA, D[co][pixel] accumulators).
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32;                 // output tile (256 pixels = 4 waves x 64)
constexpr int PH = TH + 2, PW = TW + 2;        // halo patch 10 x 34 pixels
constexpr int PPIX = PH * PW;                  // 340
constexpr int PGROUPS = (PPIX + 7) / 8;        // 43 DMA groups of 8 pixels
constexpr int PATCH_BYTES = 44 * 1024;         // 4 waves x 11 groups x 1 KB
constexpr int WSTAGE = 64 * 128;               // one tap's weights: 64 co rows x 128 B
constexpr int CPAD = 64 + 8;
constexpr int WSTAGES = 4;                     // weight ring: three taps in flight behind the one being multiplied
constexpr int LDS_BYTES = PATCH_BYTES + WSTAGES * WSTAGE;   // 77,824 B -> 2 blocks per CU

struct C64Args {
  const unsigned short* in;
  const unsigned short* w;       // [64][3][3][64]
  unsigned short* out;
  const unsigned short* addend;
  double* stat_sum;
  double* stat_sumsq;
  int stat_slots;
  int N, H, W, tiles_y, tiles_x;
};

__device__ __attribute__((aligned(256))) unsigned char g_c64_zero_page[256];

__device__ __forceinline__ float bfb(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// 16-byte-per-lane LDS-DMA in inline asm (see conv_wgrad.hip): keeps the loads out of hipcc's vmcnt bookkeeping,
// the tap loop below counts them itself.  M0 (the DMA's LDS base) is saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256) void conv3x3_c64_kernel(C64Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* patch = smem;
  unsigned char* wst = smem + PATCH_BYTES;
  const unsigned patch_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)patch;
  const unsigned wst_lds = patch_lds + PATCH_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  int bt = blockIdx.x;
  const int tx = bt % a.tiles_x; bt /= a.tiles_x;
  const int ty = bt % a.tiles_y;
  const int n = bt / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;

  // ---- stage the halo patch: DMA group g = 8 consecutive patch pixels (1 KB), lane -> (pixel 8g + l/8, slot l%8),
  //      slot p of pixel q holds global chunk p ^ (q & 7)
  const int r8 = lane >> 3, gch = (lane & 7) ^ r8;
  const unsigned char* zp = g_c64_zero_page + (lane & 7) * 16;
#pragma unroll
  for (int j = 0; j < 11; ++j) {
    const int g = wave + 4 * j;
    const int pp = g * 8 + r8;
    const int py = pp / PW, px = pp - py * PW;
    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
    const bool ok = pp < PPIX && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    const void* src = ok ? (const void*)(a.in + (((size_t)n * a.H + yy) * a.W + xx) * 64 + gch * 8) : (const void*)zp;
    glds16(src, patch_lds + g * 1024);
  }
  // weights of one tap: 64 rows x 128 B = 8 DMA groups, 2 per wave; row co = 8g + l/8, chunk (l%8) ^ (co&7) = gch
  const unsigned short* wsrc = a.w + (size_t)(wave * 8 + r8) * 576 + gch * 8;
  auto issue_w = [&](int tap) {
    const unsigned stage = wst_lds + (tap % WSTAGES) * WSTAGE;
    glds16(wsrc + tap * 64, stage + wave * 1024);
    glds16(wsrc + 32 * 576 + tap * 64, stage + (wave + 4) * 1024);
  };
  issue_w(0); issue_w(1); issue_w(2);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this lane's 4 output pixels (one per M-tile): tile row 2*wave + (i>>1), tile column (i&1)*16 + fr;
  // patch pixel of tap (0,0) = the same coordinates (the patch origin is the tile origin - 1)
  int pp0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pp0[i] = (2 * wave + (i >> 1)) * PW + (i & 1) * 16 + fr;

#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    // the patch and the weights of this tap have landed: only the (up to two) later taps' loads may be in flight
    if (tap <= 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (tap == 7) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // ... for every wave; and stage (tap-1)%4 is free again
    if (tap + 3 < 9) issue_w(tap + 3);
    const unsigned char* Bs = wst + (tap % WSTAGES) * WSTAGE;
    const int kh = tap / 3, kw = tap - kh * 3;
    const int poff = kh * PW + kw;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pp = pp0[i] + poff;
        af[i] = *reinterpret_cast<const bf16x8*>(patch + pp * 128 + (((ks * 4 + fg) ^ (pp & 7)) << 4));
        const int co = i * 16 + fr;
        bfr[i] = *reinterpret_cast<const bf16x8*>(Bs + co * 128 + (((ks * 4 + fg) ^ (co & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // D[co][pixel]
    }
  }
  __syncthreads();                                       // patch / weight stages are dead from here on

  // ---- epilogue (C tile [256 px][64 co] through LDS; pixel m = 32 * tile_row + tile_col)
  unsigned short* Cs = reinterpret_cast<unsigned short*>(smem);
  bool valid[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    valid[i] = (y0 + 2 * wave + (i >> 1) < a.H) && (x0 + (i & 1) * 16 + fr < a.W);
  auto pix_index = [&](int m) -> long long {             // flat output pixel of tile pixel m, or -1
    const int yy = y0 + (m >> 5), xx = x0 + (m & 31);
    return (yy < a.H && xx < a.W) ? (((long long)n * a.H + yy) * a.W + xx) : -1;
  };
  const int crow0 = (wave * 64 + fr) * CPAD + fg * 4;
  if (a.addend) {
    for (int idx = tid; idx < 256 * 8; idx += 256) {
      const int row = idx >> 3, ch = idx & 7;
      const long long pix = pix_index(row);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (pix >= 0) v = *reinterpret_cast<const u32x4*>(a.addend + pix * 64 + ch * 8);
      *reinterpret_cast<u32x4*>(Cs + row * CPAD + ch * 8) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(Cs + crow0 + i * 16 * CPAD + j * 16);
        acc[i][j][0] += __uint_as_float(v[0] << 16);
        acc[i][j][1] += __uint_as_float(v[0] & 0xFFFF0000u);
        acc[i][j][2] += __uint_as_float(v[1] << 16);
        acc[i][j][3] += __uint_as_float(v[1] & 0xFFFF0000u);
      }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x2 v = {0u, 0u};                                // pixels outside the image stay exact zeros (statistics!)
      if (valid[i]) {
        v[0] = (unsigned)f32_to_bf16_bits(acc[i][j][0]) | ((unsigned)f32_to_bf16_bits(acc[i][j][1]) << 16);
        v[1] = (unsigned)f32_to_bf16_bits(acc[i][j][2]) | ((unsigned)f32_to_bf16_bits(acc[i][j][3]) << 16);
      }
      *reinterpret_cast<u32x2*>(Cs + crow0 + i * 16 * CPAD + j * 16) = v;
    }
  __syncthreads();
  for (int idx = tid; idx < 256 * 8; idx += 256) {
    const int row = idx >> 3, ch = idx & 7;
    const long long pix = pix_index(row);
    if (pix >= 0)
      *reinterpret_cast<u32x4*>(a.out + pix * 64 + ch * 8) = *reinterpret_cast<const u32x4*>(Cs + row * CPAD + ch * 8);
  }
  if (a.stat_sum) {
    const int col = tid & 63, part = tid >> 6;            // 4 threads per channel, 64 rows each
    float s = 0.f, q = 0.f;
    for (int r = part * 64; r < (part + 1) * 64; ++r) {
      const float v = bfb(Cs[r * CPAD + col]);
      s += v; q += v * v;
    }
    float* red = reinterpret_cast<float*>(smem + 256 * CPAD * 2);
    red[tid] = s; red[256 + tid] = q;
    __syncthreads();
    if (tid < 64) {
      double ds = 0.0, dq = 0.0;
#pragma unroll
      for (int p = 0; p < 4; ++p) { ds += (double)red[p * 64 + tid]; dq += (double)red[256 + p * 64 + tid]; }
      const size_t slot = (size_t)(blockIdx.x % a.stat_slots) * 64 + tid;
      atomicAdd(a.stat_sum + slot, ds);
      atomicAdd(a.stat_sumsq + slot, dq);
    }
  }
}

}  // namespace

// called by isic_conv2d_igemm_bf16 for Cin = Cout = 64, 3x3, stride 1, pad 1
int isic_conv3x3_c64_launch(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int H, int W,
                            const uint16_t* addend, double* stat_sum, double* stat_sumsq, int stat_slots,
                            hipStream_t stream) {
  C64Args a;
  a.in = in; a.w = w; a.out = out; a.addend = addend;
  a.stat_sum = stat_sum; a.stat_sumsq = stat_sumsq; a.stat_slots = stat_slots > 0 ? stat_slots : 1;
  a.N = N; a.H = H; a.W = W;
  a.tiles_y = ceil_div(H, TH); a.tiles_x = ceil_div(W, TW);
  const int64_t blocks = (int64_t)N * a.tiles_y * a.tiles_x;
  if (blocks > 0x7FFFFFFFLL) return ISIC_ERR_UNSUPPORTED;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess)
      return ISIC_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(conv3x3_c64_kernel, dim3((unsigned)blocks), dim3(256), LDS_BYTES, stream, a);
  return ISIC_OK;
}
