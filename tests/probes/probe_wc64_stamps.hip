// Phase timestamps of the all-taps 64 -> 64 weight-gradient kernel (development probe, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -Iinclude -Imultimodal-isic_amd/csrc tests/probes/probe_wc64_stamps.hip -o tests/probes/build/wc64_stamps.so
#define WC64_STAMPS 1
#include "../../multimodal-isic_amd/csrc/conv_wgrad_c64.hip"

extern "C" int probe_wc64_run(const uint16_t* x, const uint16_t* dy, float* dw, int N, int H, int W, void* ws,
                              unsigned long long* stamps_host) {
  int rc = isic_wgrad_c64_launch(x, dy, dw, N, H, W, ws, nullptr);
  if (rc) return rc;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(stamps_host, HIP_SYMBOL(g_wc64_stamps), sizeof(unsigned long long) * 256 * 64 * 4) == hipSuccess ? 0 : -2;
}
extern "C" size_t probe_wc64_ws(int N, int H, int W) { return isic_wgrad_c64_workspace_bytes(N, H, W); }
