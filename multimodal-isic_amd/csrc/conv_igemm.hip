// Implicit-GEMM convolution on bf16 MFMA (gfx950), NHWC activations, [Cout][Kh][Kw][Cin] weights.
//
//   out[m, co] = sum_{kh,kw,ci} in[pix(m,kh,kw), ci] * w[co, kh, kw, ci]      m = (n, ho, wo)
//
// GEMM view: M = N*Hout*Wout (pixels), N = Cout, K = Kh*Kw*Cin with BOTH operands K-contiguous
// (NHWC rows are channel-contiguous, weights are stored K-major), which is exactly the operand
// shape v_mfma_f32_16x16x32_bf16 wants: 8 consecutive k per lane = one 16-byte load.
//
// The same kernel is the forward pass (up = stride, down = 1) and the data-gradient pass
// (up = 1, down = stride, flipped/transposed weights): the source pixel of tap (kh,kw) for output
// (ho,wo) is ((ho*up + kh - pad)/down, (wo*up + kw - pad)/down) and taps whose source is fractional
// or outside the image contribute zero.
//
// The reference has no convolution kernel of its own (its encoder is an un-vendored ConvMAE run
// through torch, save_latent.py:42-60); BASELINE.json configs[1] names ResNet-18, whose layer table
// is SURVEY.md 8d.
//
// Structure: 256 threads = 4 waves; block tile BM x BN (256x64 as 4x1 waves for Cout = 64,
// 128x128 as 2x2 waves otherwise), each wave a 64x64 sub-tile = 4x4 MFMA tiles (64 accumulator
// VGPRs); BK = 64 = one (kh,kw) tap x 64 channels.  Global -> registers -> LDS staging with the next
// K-tile's loads issued before the current tile's MFMAs; LDS rows are 128 B with a 16-byte-chunk XOR
// swizzle (chunk ^= row & 7) so the ds_read_b128 fragment reads are conflict-free; the epilogue
// transposes through LDS so every global store is a full 16-byte-per-lane row segment.
#include "common.h"

namespace {

constexpr int BK = 64;  // bf16 elements per K-tile (128 bytes per LDS row)

struct ConvArgs {
  const unsigned short* in;
  const unsigned short* w;
  unsigned short* out;
  int N, Hin, Win, Cin, Hout, Wout, Cout, Kh, Kw, up, down_shift, pad;
  int M;        // N*Hout*Wout
  int Ktiles;   // Kh*Kw*Cin/64
  int ctiles;   // Cin/64
};

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(BM == WAVES_M * 64 && BN == WAVES_N * 64, "64x64 per wave");
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int CPAD = BN + 8;  // epilogue row stride (elements)
  constexpr int C_BYTES = BM * CPAD * 2;
  constexpr int LDS_BYTES = (A_BYTES + B_BYTES) > C_BYTES ? (A_BYTES + B_BYTES) : C_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  unsigned char* As = smem;
  unsigned char* Bs = smem + A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  // ---- staging assignment: 16-byte chunk `sc` of rows `sr + 32*i`
  constexpr int AROWS = BM / 32, BROWS = BN / 32;
  const int sr = tid >> 3, sc = tid & 7;
  int a_base[AROWS];   // element offset of pixel (n, 0, 0) channel 0 ; -1 if row >= M
  int a_hb[AROWS], a_wb[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + sr + 32 * i;
    if (m < a.M) {
      const int hw = a.Hout * a.Wout;
      const int n = m / hw, r = m - n * hw;
      const int ho = r / a.Wout, wo = r - ho * a.Wout;
      a_base[i] = n * a.Hin * a.Win;
      a_hb[i] = ho * a.up - a.pad;
      a_wb[i] = wo * a.up - a.pad;
    } else {
      a_base[i] = -1; a_hb[i] = 0; a_wb[i] = 0;
    }
  }
  const int dmask = (1 << a.down_shift) - 1;
  const size_t Ktot = (size_t)a.Kh * a.Kw * a.Cin;

  u32x4 ra[AROWS], rb[BROWS];
  auto gload = [&](int kt) {
    const int tap = kt / a.ctiles, c0 = (kt - tap * a.ctiles) * BK;
    const int kh = tap / a.Kw, kw = tap - kh * a.Kw;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int hn = a_hb[i] + kh, wn_ = a_wb[i] + kw;
      const int hi = hn >> a.down_shift, wi = wn_ >> a.down_shift;
      const bool ok = a_base[i] >= 0 && hn >= 0 && wn_ >= 0 && ((hn | wn_) & dmask) == 0 && hi < a.Hin && wi < a.Win;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) {
        const size_t off = ((size_t)(a_base[i] + hi * a.Win + wi)) * a.Cin + c0 + sc * 8;
        v = *reinterpret_cast<const u32x4*>(a.in + off);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const int co = n0 + sr + 32 * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (co < a.Cout) v = *reinterpret_cast<const u32x4*>(a.w + (size_t)co * Ktot + (size_t)kt * BK + sc * 8);
      rb[i] = v;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int r = sr + 32 * i;
      *reinterpret_cast<u32x4*>(As + r * 128 + ((sc ^ (r & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const int r = sr + 32 * i;
      *reinterpret_cast<u32x4*>(Bs + r * 128 + ((sc ^ (r & 7)) << 4)) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  gload(0);
  for (int kt = 0; kt < a.Ktiles; ++kt) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (kt + 1 < a.Ktiles) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra_ = wm * 64 + i * 16 + fr;
        af[i] = *reinterpret_cast<const bf16x8*>(As + ra_ * 128 + (((ks * 4 + fg) ^ (ra_ & 7)) << 4));
        const int rb_ = wn * 64 + i * 16 + fr;
        bfr[i] = *reinterpret_cast<const bf16x8*>(Bs + rb_ * 128 + (((ks * 4 + fg) ^ (rb_ & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: accumulators -> bf16 -> LDS [BM][CPAD] -> 16-byte row segments to global
  __syncthreads();
  unsigned short* Cs = reinterpret_cast<unsigned short*>(smem);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + i * 16 + fg * 4 + r;
        const int col = wn * 64 + j * 16 + fr;
        Cs[row * CPAD + col] = f32_to_bf16_bits(acc[i][j][r]);
      }
  __syncthreads();
  constexpr int CHUNKS = BN / 8;  // 16-byte chunks per output row
  for (int idx = tid; idx < BM * CHUNKS; idx += 256) {
    const int row = idx / CHUNKS, ch = idx - row * CHUNKS;
    const int m = m0 + row;
    if (m < a.M) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(Cs + row * CPAD + ch * 8);
      *reinterpret_cast<u32x4*>(a.out + (size_t)m * a.Cout + n0 + ch * 8) = v;
    }
  }
}

// master fp32 weights in [O][Kh][Kw][I] memory order (torch channels_last of an OIHW tensor) ->
// bf16 forward copy (same order) and bf16 dgrad copy [I][Kh][Kw][O] with both taps flipped.
__global__ void weight_prep_kernel(const float* __restrict__ w, unsigned short* __restrict__ w_fwd,
                                   unsigned short* __restrict__ w_dgrad, int O, int I, int Kh, int Kw) {
  const int64_t n = (int64_t)O * Kh * Kw * I;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % I);
    int64_t t = idx / I;
    const int kw = (int)(t % Kw); t /= Kw;
    const int kh = (int)(t % Kh);
    const int o = (int)(t / Kh);
    const unsigned short b = f32_to_bf16_bits(w[idx]);
    if (w_fwd) w_fwd[idx] = b;
    if (w_dgrad) w_dgrad[(((int64_t)i * Kh + (Kh - 1 - kh)) * Kw + (Kw - 1 - kw)) * O + o] = b;
  }
}

}  // namespace

extern "C" {

int isic_conv2d_igemm_bf16(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int Hin, int Win, int Cin,
                           int Hout, int Wout, int Cout, int Kh, int Kw, int up, int down, int pad, void* stream) {
  ISIC_CHECK_ARG(in && w && out);
  ISIC_CHECK_ARG(N > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Kh > 0 && Kw > 0 && up > 0);
  ISIC_CHECK_ARG(down == 1 || down == 2 || down == 4);
  if (Cin % 64 != 0 || Cout % 64 != 0) return ISIC_ERR_UNSUPPORTED;
  const int64_t M64 = (int64_t)N * Hout * Wout;
  if (M64 > 0x7FFFFFFFLL || (int64_t)N * Hin * Win > 0x7FFFFFFFLL) return ISIC_ERR_UNSUPPORTED;
  ConvArgs a;
  a.in = in; a.w = w; a.out = out;
  a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
  a.Kh = Kh; a.Kw = Kw; a.up = up; a.down_shift = down == 1 ? 0 : (down == 2 ? 1 : 2); a.pad = pad;
  a.M = (int)M64; a.ctiles = Cin / 64; a.Ktiles = Kh * Kw * a.ctiles;
  if (Cout % 128 != 0) {
    dim3 grid(ceil_div(a.M, 256), Cout / 64);
    hipLaunchKernelGGL((conv_igemm_kernel<256, 64, 4, 1>), grid, dim3(256), 0, as_stream(stream), a);
  } else {
    dim3 grid(ceil_div(a.M, 128), Cout / 128);
    hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2>), grid, dim3(256), 0, as_stream(stream), a);
  }
  return isic_launch_status();
}

int isic_conv_weight_prep_bf16(const float* w_krsc, uint16_t* w_fwd, uint16_t* w_dgrad, int O, int I, int Kh, int Kw,
                               void* stream) {
  ISIC_CHECK_ARG(w_krsc && (w_fwd || w_dgrad) && O > 0 && I > 0 && Kh > 0 && Kw > 0);
  const int64_t n = (int64_t)O * I * Kh * Kw;
  int64_t grid = (n + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(weight_prep_kernel, dim3((int)grid), dim3(256), 0, as_stream(stream), w_krsc, w_fwd, w_dgrad, O, I,
                     Kh, Kw);
  return isic_launch_status();
}

}  // extern "C"
