import os, sys
sys.path.insert(0, "multimodal-isic_amd")
import torch
from isic_hip.lib import call
DEV = "cuda:0"; BF = torch.bfloat16
N, H, W = 2, 16, 16
g = torch.Generator().manual_seed(11)
x = torch.randn(N, H, W, 64, generator=g).to(DEV).to(BF)
wf = (torch.randn(64, 3, 3, 64, generator=g) / 24).to(DEV).to(BF)
def run(v):
    call("isic_debug_set_conv_variant", v * 10 + 2)
    out = torch.zeros_like(x)
    call("isic_conv2d_igemm_bf16", x, wf, out, N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, None, None, None, 0)
    torch.cuda.synchronize()
    return out
ref = run(1)
for rep in range(3):
    got = run(3)
    bad = (ref != got)
    print("bad", int(bad.sum()), "of", bad.numel())
    idx = bad.nonzero()
    if len(idx):
        print("n", idx[:, 0].unique().tolist(), "y", idx[:, 1].unique().tolist(), "x", idx[:, 2].unique().tolist())
        print("c", idx[:, 3].unique().tolist())
        print(idx[:10].tolist())
        print(ref[bad][:8].tolist(), got[bad][:8].tolist())
