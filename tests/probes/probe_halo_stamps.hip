// Per-tile timestamps of the pixels-staged-once convolution (development probe, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tests/probes/probe_halo_stamps.hip -o tests/probes/build/halo_stamps.so
#define HALO_STAMPS 1
#include "../../multimodal-isic_amd/csrc/conv_halo.hip"

extern "C" int probe_halo_run(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int H, int C, const uint16_t* addend,
                              double* s0, double* s1, unsigned long long* stamps_host) {
  static unsigned long long zeros[256 * 64 * 4];
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_halo_stamps), zeros, sizeof(zeros)) != hipSuccess) return -3;   // no stale tiles
  int rc = isic_conv_halo_launch(in, w, out, N, H, H, C, C, addend, s0, s1, 32, nullptr, nullptr, nullptr);
  if (rc) return rc;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(stamps_host, HIP_SYMBOL(g_halo_stamps), sizeof(unsigned long long) * 256 * 64 * 4) == hipSuccess ? 0 : -2;
}
