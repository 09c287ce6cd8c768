// Phase timestamps of the halo-resident 64 -> 64 kernel (development probe, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -Iinclude -Imultimodal-isic_amd/csrc tests/probes/probe_c64_stamps.hip -o tests/probes/build/c64_stamps.so
#define C64_STAMPS 1
#include "../../multimodal-isic_amd/csrc/conv_c64.hip"

extern "C" int probe_c64_run(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int H, int W, double* ssum,
                             double* ssq, int slots, unsigned long long* stamps_host) {
  int rc = isic_conv3x3_c64_launch(1, in, w, out, N, H, W, nullptr, ssum, ssq, slots, nullptr);
  if (rc) return rc;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(stamps_host, HIP_SYMBOL(g_c64_stamps), sizeof(unsigned long long) * 8192 * 4) == hipSuccess ? 0 : -2;
}

extern "C" int probe_c64p_run(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int H, int W, double* ssum,
                              double* ssq, int slots, unsigned long long* stamps_host) {
  int rc = isic_conv3x3_c64_launch(2, in, w, out, N, H, W, nullptr, ssum, ssq, slots, nullptr);
  if (rc) return rc;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(stamps_host, HIP_SYMBOL(g_c64p_stamps), sizeof(unsigned long long) * 256 * 32 * 4) == hipSuccess ? 0 : -2;
}
