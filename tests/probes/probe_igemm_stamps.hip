// Per-K-tile timestamps of the staging-wave implicit GEMM (development probe, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -Iinclude -Imultimodal-isic_amd/csrc tests/probes/probe_igemm_stamps.hip -o tests/probes/build/igemm_stamps.so
#define IGEMM_STAMPS 1
#include "../../include/isic_hip_test.h"
#include "../../multimodal-isic_amd/csrc/conv_igemm.hip"
// the 64 -> 64 halo kernels are not part of this probe
int isic_conv3x3_c64_launch(int, const uint16_t*, const uint16_t*, uint16_t*, int, int, int, const uint16_t*, double*, double*, int, hipStream_t) { return ISIC_ERR_UNSUPPORTED; }
// ... nor is the pixels-staged-once kernel
bool isic_conv_halo_supported(int, int, int, int, int) { return false; }
int isic_conv_halo_launch(const uint16_t*, const uint16_t*, uint16_t*, int, int, int, int, int, const uint16_t*, double*, double*, int, hipStream_t) { return ISIC_ERR_UNSUPPORTED; }

extern "C" int probe_igemm_run(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int H, int C,
                               unsigned long long* stamps_host, int mode) {
  // staging MODE pinned per call (units digit = MODE + 1), generic kernel forced for every shape (tens 1, hundreds 1)
  int rc = isic_test_conv2d_igemm_variant_bf16(in, w, out, N, H, H, C, H, H, C, 3, 3, 1, 1, 1, nullptr, nullptr, nullptr, 0,
                                               110 + mode + 1, nullptr);
  if (rc) return rc;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(stamps_host, HIP_SYMBOL(g_igemm_stamps), sizeof(unsigned long long) * 256 * 40 * 8) == hipSuccess ? 0 : -2;
}
