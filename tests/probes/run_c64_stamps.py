"""Driver of probe_c64_stamps.hip: prints the median phase lengths (shader clocks) of the blocks."""
import ctypes, os, sys
import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
so = ctypes.CDLL(os.path.join(here, "build", sys.argv[1] if len(sys.argv) > 1 else "c64_stamps.so"))
N, H, W = 512, 56, 56
dev = "cuda:0"
x = torch.randn(N, H, W, 64, device=dev).bfloat16()
w = (torch.randn(64, 3, 3, 64, device=dev) / 24).bfloat16()
out = torch.empty_like(x)
st = torch.zeros(2, 32, 64, device=dev, dtype=torch.float64)
stamps = np.zeros((8192, 4), dtype=np.uint64)
P = ctypes.c_void_p
for it in range(3):
    rc = so.probe_c64_run(P(x.data_ptr()), P(w.data_ptr()), P(out.data_ptr()), N, H, W, P(st[0].data_ptr()),
                          P(st[1].data_ptr()), 32, stamps.ctypes.data_as(P))
    assert rc == 0, rc
s = stamps.astype(np.int64)
nb = min(8192, N * 7 * 2)
s = s[:nb]
d = np.diff(s, axis=1)
print("blocks", nb, "span (clk)", int(s[:, 3].max() - s[:, 0].min()))
for k, name in enumerate(["prologue (patch landed)", "nine taps", "epilogue"]):
    print(f"{name:26s} median {int(np.median(d[:, k])):7d}  p10 {int(np.percentile(d[:, k], 10)):7d}  p90 {int(np.percentile(d[:, k], 90)):7d}")
print("block lifetime median", int(np.median(s[:, 3] - s[:, 0])))

# persistent kernel: per tile stamps of wave 0
for label, ss, sq in (("persistent, no stats", None, None), ("persistent, stats", P(st[0].data_ptr()), P(st[1].data_ptr()))):
    ps = np.zeros((256, 32, 4), dtype=np.uint64)
    for it in range(3):
        rc = so.probe_c64p_run(P(x.data_ptr()), P(w.data_ptr()), P(out.data_ptr()), N, H, W, ss, sq, 32, ps.ctypes.data_as(P))
        assert rc == 0, rc
    p = ps.astype(np.int64)[:, :28, :]
    wait = p[:, :, 1] - p[:, :, 0]
    comp = p[:, :, 2] - p[:, :, 1]
    epi = p[:, :, 3] - p[:, :, 2]
    per = np.diff(p[:, :, 0], axis=1)
    print(label)
    print("  tile period median", int(np.median(per)), " wait+barrier", int(np.median(wait[:, 1:])), " issue DMA + 144 MFMA", int(np.median(comp)),
          " epilogue", int(np.median(epi)), " first-tile wait", int(np.median(wait[:, 0])))
    print("  block total median", int(np.median(p[:, 27, 3] - p[:, 0, 0])))
