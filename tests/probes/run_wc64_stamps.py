"""Driver of probe_wc64_stamps.hip: median phase lengths (shader clocks) per tile."""
import ctypes, os
import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
import sys
so = ctypes.CDLL(os.path.join(here, "build", sys.argv[1] if len(sys.argv) > 1 else "wc64_stamps.so"))
so.probe_wc64_ws.restype = ctypes.c_size_t
N, H, W = 512, 56, 56
dev = "cuda:0"
x = torch.randn(N, H, W, 64, device=dev).bfloat16()
dy = torch.randn(N, H, W, 64, device=dev).bfloat16()
dw = torch.zeros(64 * 9 * 64, device=dev)
ws = torch.empty(so.probe_wc64_ws(N, H, W), device=dev, dtype=torch.uint8)
P = ctypes.c_void_p
st = np.zeros((256, 64, 4), dtype=np.uint64)
for it in range(3):
    rc = so.probe_wc64_run(P(x.data_ptr()), P(dy.data_ptr()), P(dw.data_ptr()), N, H, W, P(ws.data_ptr()), st.ctypes.data_as(P))
    assert rc == 0, rc
p = st.astype(np.int64)[:, :49, :]
print("tile period median", int(np.median(np.diff(p[:, :, 0], axis=1))), " wait+barrier", int(np.median((p[:, 1:, 1] - p[:, 1:, 0]))),
      " body (12 DMA + 144 MFMA)", int(np.median(p[:, :, 2] - p[:, :, 1])), " first wait", int(np.median(p[:, 0, 1] - p[:, 0, 0])))
print("vmcnt wait", int(np.median(p[:, 1:, 3] - p[:, 1:, 0])), " barrier wait", int(np.median(p[:, 1:, 1] - p[:, 1:, 3])))
print("block total median", int(np.median(p[:, 48, 2] - p[:, 0, 0])))
