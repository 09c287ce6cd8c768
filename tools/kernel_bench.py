"""Per-kernel micro-benchmark of the encoder kernels at the BASELINE.json configs[1] shapes
(N = 512 images of 224x224 per step): HIP-event timing of each ResNet-18 convolution shape
(forward / data-gradient / weight-gradient) and of the BatchNorm passes.  Developer tool:
    python tools/kernel_bench.py [--n 512] [--iters 5]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-isic_amd"))
import torch
from isic_hip.lib import call

BF = torch.bfloat16
DEV = "cuda:0"
# (name, Cin, Cout, k, stride, pad, Hin) ; count = how many such convs in ResNet-18
SHAPES = [("l1 3x3 64->64 @56", 64, 64, 3, 1, 1, 56, 4), ("l2.0 3x3/2 64->128 @56", 64, 128, 3, 2, 1, 56, 1),
          ("l2 3x3 128->128 @28", 128, 128, 3, 1, 1, 28, 3), ("l2.ds 1x1/2 64->128", 64, 128, 1, 2, 0, 56, 1),
          ("l3.0 3x3/2 128->256 @28", 128, 256, 3, 2, 1, 28, 1), ("l3 3x3 256->256 @14", 256, 256, 3, 1, 1, 14, 3),
          ("l3.ds 1x1/2 128->256", 128, 256, 1, 2, 0, 28, 1), ("l4.0 3x3/2 256->512 @14", 256, 512, 3, 2, 1, 14, 1),
          ("l4 3x3 512->512 @7", 512, 512, 3, 1, 1, 7, 3), ("l4.ds 1x1/2 256->512", 256, 512, 1, 2, 0, 14, 1)]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--variant", type=int, default=0,
                    help="kernel choice pinned per call (include/isic_hip_test.h); 0 = the shipped dispatch")
    ap.add_argument("--only", default="", help="substring filter on the conv shape names; also skips the extra kernels")
    a = ap.parse_args()
    N = a.n
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'conv':28s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF/s':>7s} | {'dgrad ms':>8s} {'TF/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s}")
    for name, ci, co, k, s, p, h, cnt in SHAPES:
        if a.only and a.only not in name:
            continue
        ho = (h + 2 * p - k) // s + 1
        x = torch.randn(N, h, h, ci, device=DEV).to(BF)
        dy = torch.randn(N, ho, ho, co, device=DEV).to(BF)
        w = (torch.randn(co, ci, k, k, device=DEV) / (ci * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
        wf = torch.empty(co * ci * k * k, device=DEV, dtype=BF)
        wd = torch.empty_like(wf)
        call("isic_conv_weight_prep_bf16", w, wf, wd, co, ci, k, k)
        out = torch.empty(N, ho, ho, co, device=DEV, dtype=BF)
        dx = torch.empty(N, h, h, ci, device=DEV, dtype=BF)
        dw = torch.zeros_like(w)
        acc = torch.zeros(2, 32, co, device=DEV, dtype=torch.float64)
        gf = 2.0 * N * ho * ho * co * k * k * ci / 1e9
        tf = timeit(lambda: call("isic_test_conv2d_igemm_variant_bf16", x, wf, out, N, h, h, ci, ho, ho, co, k, k, s, 1, p, None, acc[0], acc[1], 32, a.variant), a.iters)
        td = timeit(lambda: call("isic_test_conv2d_igemm_variant_bf16", dy, wd, dx, N, ho, ho, co, h, h, ci, k, k, 1, s, k - 1 - p, None, None, None, 0, a.variant), a.iters)
        wsb = torch.empty(call("isic_conv2d_wgrad_workspace_bytes", N, ci, ho, ho, co, k, k), device=DEV, dtype=torch.uint8)
        tw = timeit(lambda: call("isic_conv2d_wgrad_bf16", x, dy, dw, N, h, h, ci, ho, ho, co, k, k, s, p, wsb, wsb.numel()), a.iters)
        print(f"{name:28s} {gf:8.1f} | {tf:8.3f} {gf / tf:7.0f} | {td:8.3f} {gf / td:7.0f} | {tw:8.3f} {gf / tw:7.0f}   x{cnt}")
        tot["fwd"] += tf * cnt; tot["dgrad"] += td * cnt; tot["wgrad"] += tw * cnt
    print("per-step totals (ms):", {k: round(v, 2) for k, v in tot.items()}, "sum", round(sum(tot.values()), 2))
    if a.only:
        return
    # stem
    x4 = torch.randn(N, 224, 224, 4, device=DEV).to(BF)
    ws = torch.randn(64 * 7 * 8 * 4, device=DEV).to(BF)
    so = torch.empty(N, 112, 112, 64, device=DEV, dtype=BF)
    sdw = torch.zeros(64, 3, 7, 7, device=DEV).contiguous(memory_format=torch.channels_last)
    t1 = timeit(lambda: call("isic_conv_stem_fwd_bf16", x4, ws, so, N, 224, 224, 112, 112), a.iters)
    wsp = torch.empty(call("isic_conv_stem_wgrad_workspace_bytes"), device=DEV, dtype=torch.uint8)
    t2 = timeit(lambda: call("isic_conv_stem_wgrad_bf16", x4, so, sdw, N, 224, 224, 112, 112, wsp, wsp.numel()), a.iters)
    gfs = 2.0 * N * 112 * 112 * 64 * 147 / 1e9
    print(f"stem fwd {t1:.3f} ms ({gfs / t1:.0f} TF/s alg)  stem wgrad {t2:.3f} ms ({gfs / t2:.0f} TF/s alg)")
    # BN passes on the layer1 activation (N x 56 x 56 x 64)
    rows, C = N * 56 * 56, 64
    xa = torch.randn(rows, C, device=DEV).to(BF)
    ya, dya, dxa = torch.empty_like(xa), torch.randn(rows, C, device=DEV).to(BF), torch.empty_like(xa)
    accb = torch.zeros(2, 1, C, device=DEV, dtype=torch.float64)
    sc, sh, mean, rstd = (torch.ones(C, device=DEV) for _ in range(4))
    gam = torch.ones(C, device=DEV)
    gb = rows * C * 2 / 1e9
    for nm, fn, passes in (
        ("bn_stats", lambda: call("isic_bn_stats_bf16", xa, rows, C, accb[0], accb[1]), 1),
        ("bn_apply", lambda: call("isic_bn_apply_bf16", xa, sc, sh, None, ya, rows, C, 1), 2),
        ("bn_apply+res", lambda: call("isic_bn_apply_bf16", xa, sc, sh, dya, ya, rows, C, 1), 3),
        ("bn_bwd_reduce(y)", lambda: call("isic_bn_bwd_reduce_bf16", dya, xa, ya, mean, rstd, rows, C, 1, None, None, accb[0], accb[1]), 3),
        ("bn_bwd_reduce(x)", lambda: call("isic_bn_bwd_reduce_bf16", dya, xa, None, mean, rstd, rows, C, 1, sc, sh, accb[0], accb[1]), 2),
        ("bn_bwd_apply(y)+res", lambda: call("isic_bn_bwd_apply_bf16", dya, xa, ya, mean, rstd, gam, accb[0], accb[1], rows, C, 1, None, None, dxa, ya, None, None), 5),
        ("bn_bwd_apply(x)", lambda: call("isic_bn_bwd_apply_bf16", dya, xa, None, mean, rstd, gam, accb[0], accb[1], rows, C, 1, sc, sh, dxa, None, None, None), 3),
    ):
        t = timeit(fn, a.iters)
        print(f"{nm:22s} {t:7.3f} ms  {passes * gb / t * 1e3 / 1e3:7.2f} TB/s ({passes} passes of {gb * 1e3:.0f} MB)")


if __name__ == "__main__":
    main()
