"""Driver of probe_igemm_stamps.hip: median per-K-tile phase lengths (s_memtime ticks) of the 128x128 staging-wave kernel."""
import ctypes, os
import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
so = ctypes.CDLL(os.path.join(here, "build", "igemm_stamps.so"))
N, H, C = 512, 14, 256
dev = "cuda:0"
x = torch.randn(N, H, H, C, device=dev).bfloat16()
w = (torch.randn(C, 3, 3, C, device=dev) / 48).bfloat16()
out = torch.empty_like(x)
P = ctypes.c_void_p
import sys
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
st = np.zeros((256, 40, 8), dtype=np.uint64)
for it in range(3):
    rc = so.probe_igemm_run(P(x.data_ptr()), P(w.data_ptr()), P(out.data_ptr()), N, H, C, st.ctypes.data_as(P), mode)
    assert rc == 0, rc
print("mode", mode)
p = st.astype(np.int64)[:, 2:34, :]
med = lambda a: int(np.median(a))
print("K-tile period (MFMA wave)", med(np.diff(p[:, :, 0], axis=1)))
print("MFMA wave : barrier wait", med(p[:, :, 1] - p[:, :, 0]), " compute", med(p[:, :, 2] - p[:, :, 1]))
print("stager    : vmcnt wait", med(p[:, :, 4] - p[:, :, 3]), " barrier wait", med(p[:, :, 5] - p[:, :, 4]), " issue 8 DMA", med(p[:, :, 6] - p[:, :, 5]))
