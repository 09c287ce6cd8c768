"""Driver of probe_halo_stamps.hip: where a tile of conv_halo_kernel spends its time (s_memtime ticks, 100 MHz -> x24 for
shader cycles at 2.4 GHz): set-up, K loop, epilogue."""
import ctypes, os, sys
import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
so = ctypes.CDLL(os.path.join(here, "build", "halo_stamps.so"))
P = ctypes.c_void_p
dev = "cuda:0"
for (N, H, C) in ((2048, 28, 128), (2048, 14, 256), (2048, 7, 512)):
    x = torch.randn(N, H, H, C, device=dev).bfloat16()
    w = (torch.randn(C, 3, 3, C, device=dev) / (C * 9) ** 0.5).bfloat16()
    out = torch.empty_like(x)
    add = torch.randn(N, H, H, C, device=dev).bfloat16()
    st = torch.zeros(2, 32, C, device=dev, dtype=torch.float64)
    for label, ad, s0, s1 in (("plain", None, None, None), ("addend", add, None, None), ("stats", None, st[0], st[1])):
        buf = np.zeros((256, 64, 4), dtype=np.uint64)
        for it in range(2):
            rc = so.probe_halo_run(P(x.data_ptr()), P(w.data_ptr()), P(out.data_ptr()), N, H, C,
                                   P(ad.data_ptr()) if ad is not None else None,
                                   P(s0.data_ptr()) if s0 is not None else None, P(s1.data_ptr()) if s1 is not None else None,
                                   buf.ctypes.data_as(P))
            assert rc == 0, rc
        p = buf.astype(np.int64)
        ntl = int((p[0, :, 0] > 0).sum())
        p = p[:, 1:ntl - 1, :]
        kloop = p[:, :, 1] - p[:, :, 0]
        epi = p[:, :, 2] - p[:, :, 1]
        setup = p[:, 1:, 0] - p[:, :-1, 2]
        period = np.diff(p[:, :, 0], axis=1)
        med = lambda a: float(np.median(a))
        print(f"C={C} H={H} {label:7s}: tiles/block {ntl}  period {med(period):7.0f}  K loop {med(kloop):7.0f}  epilogue {med(epi):6.0f}  "
              f"set-up {med(setup):5.0f}  (ticks; K-tiles {C // 64 * 9})")
