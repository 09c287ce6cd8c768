"""A/B of kernel-internal experiments of the 3x3 convolution kernels (include/isic_hip_test.h, ten-thousands digit of
`variant`) inside ONE process: bit-equality with the shipped kernel + HIP-event timing.  Developer tool:
    python tools/halo_ab.py [--n 2048] [--exps 0,1,2,3] [--layers l2,l3,l4]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-isic_amd"))
import torch
from isic_hip.lib import IsicHipError, call

BF, DEV = torch.bfloat16, "cuda:0"
LAYERS = {"l1": (64, 56), "l2": (128, 28), "l3": (256, 14), "l4": (512, 7)}


def interleaved(fns, iters, rounds=5, warm=20):
    """median over `rounds` of the per-call time of every variant, the variants timed in turn inside each round (the chip's
    clock drifts by 10 % over the first seconds of load: back-to-back blocks of one variant after another are not comparable)"""
    for f in fns:
        for _ in range(warm):
            f()
    res = [[] for _ in fns]
    for _ in range(rounds):
        for i, f in enumerate(fns):
            res[i].append(timeit(f, iters))
    return [sorted(r)[len(r) // 2] for r in res]


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


def pair_ab(a):
    N = a.n
    for name, C, Co, H in (("l2.0", 64, 128, 56), ("l3.0", 128, 256, 28), ("l4.0", 256, 512, 14)):
        Ho = H // 2
        dy = torch.randn(N, Ho, Ho, Co, device=DEV).to(BF)
        dy2 = torch.randn(N, Ho, Ho, Co, device=DEV).to(BF)
        w3 = (torch.randn(Co, C, 3, 3, device=DEV) / (C * 9) ** 0.5).contiguous(memory_format=torch.channels_last)
        w1 = (torch.randn(Co, C, 1, 1, device=DEV) / C ** 0.5).contiguous(memory_format=torch.channels_last)
        wd3, wd1 = torch.empty(Co * C * 9, device=DEV, dtype=BF), torch.empty(Co * C, device=DEV, dtype=BF)
        call("isic_conv_weight_prep_bf16", w3, None, wd3, Co, C, 3, 3)
        call("isic_conv_weight_prep_bf16", w1, None, wd1, Co, C, 1, 1)
        gf = 2.0 * N * Ho * Ho * Co * C * 10 / 1e9
        variants = [0, 2000]
        outs = [torch.zeros(N, H, H, C, device=DEV, dtype=BF) for _ in variants]
        fns = [(lambda v=v, o=o: call("isic_test_conv2d_dgrad_pair_variant_bf16", dy, wd3, dy2, wd1, o, N, Ho, Ho, Co, H, H, C, v))
               for v, o in zip(variants, outs)]
        for f in fns:
            f()
        torch.cuda.synchronize()
        ts = interleaved(fns, a.iters)
        for v, o, t in zip(variants, outs, ts):
            same = "bit-equal" if torch.equal(o, outs[0]) else f"differs max {float((o.float() - outs[0].float()).abs().max()):.3e}"
            print(f"{name} pair dgrad variant {v:4d}: {t:7.4f} ms  {gf / t:7.0f} TF/s alg  {same}", flush=True)


def wgrad_ab(a):
    N = a.n
    variants = [int(v) for v in a.exps.split(",")]
    for name in a.layers.split(","):
        C, h = LAYERS[name]
        x = torch.randn(N, h, h, C, device=DEV).to(BF)
        dy = torch.randn(N, h, h, C, device=DEV).to(BF)
        gf = 2.0 * N * h * h * C * 9 * C / 1e9
        wsb = torch.empty(call("isic_conv2d_wgrad_workspace_bytes", N, C, h, h, C, 3, 3), device=DEV, dtype=torch.uint8)
        dws = [torch.zeros(C, C, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last) for _ in variants]
        fns = [(lambda v=v, dw=dw: call("isic_test_conv2d_wgrad_variant_bf16", x, dy, dw, N, h, h, C, h, h, C, 3, 3, 1, 1, wsb,
                                        wsb.numel(), v)) for v, dw in zip(variants, dws)]
        for f in fns:
            f()
        torch.cuda.synchronize()
        ts = interleaved(fns, a.iters)
        for v, dw, t in zip(variants, dws, ts):
            same = "bit-equal" if torch.equal(dw, dws[0]) else f"differs max {float((dw - dws[0]).abs().max()):.3e}"
            print(f"{name} wgrad variant {v:2d}: {t:7.4f} ms  {gf / t:7.0f} TF/s  {same}", flush=True)


def wgrad_s2_ab(a):
    """the strided all-taps weight gradient (variant 0) against the per-tap kernel it replaces (variant 16)"""
    N = a.n
    for name, C, Co, H in (("l2.0", 64, 128, 56), ("l3.0", 128, 256, 28), ("l4.0", 256, 512, 14)):
        Ho = H // 2
        x = torch.randn(N, H, H, C, device=DEV).to(BF)
        dy = torch.randn(N, Ho, Ho, Co, device=DEV).to(BF)
        gf = 2.0 * N * Ho * Ho * Co * C * 9 / 1e9
        wsb = torch.empty(call("isic_conv2d_wgrad_workspace_bytes", N, C, Ho, Ho, Co, 3, 3), device=DEV, dtype=torch.uint8)
        variants = [32, 16]                          # 32: the all-taps kernel whatever the channel counts
        dws = [torch.zeros(Co, C, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last) for _ in variants]
        fns = [(lambda v=v, dw=dw: call("isic_test_conv2d_wgrad_variant_bf16", x, dy, dw, N, H, H, C, Ho, Ho, Co, 3, 3, 2, 1, wsb,
                                        wsb.numel(), v)) for v, dw in zip(variants, dws)]
        for f in fns:
            f()
        torch.cuda.synchronize()
        ref = float(dws[1].abs().max())
        ts = interleaved(fns, a.iters)
        for v, dw, t in zip(variants, dws, ts):
            print(f"{name} strided wgrad variant {v:2d}: {t:7.4f} ms  {gf / t:7.0f} TF/s  max |diff| / max "
                  f"{float((dw - dws[1]).abs().max()) / ref:.2e}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--exps", default="0,1", help="conv: 0 = shipped kernel, 1 = the round-3 K loop; --wgrad: variant bits")
    ap.add_argument("--layers", default="l2,l3,l4")
    ap.add_argument("--addend", action="store_true", help="also the data gradient with a residual-gradient addend")
    ap.add_argument("--pair", action="store_true", help="A/B of the downsample-block pair data gradient: variants 0 and 2000")
    ap.add_argument("--wgrad-s2", action="store_true", help="the strided all-taps weight gradient against the per-tap kernel")
    ap.add_argument("--wgrad", action="store_true", help="A/B of the weight-gradient block orders instead (isic_test_conv2d_wgrad_variant_bf16)")
    a = ap.parse_args()
    if a.wgrad_s2:
        return wgrad_s2_ab(a)
    if a.wgrad:
        return wgrad_ab(a)
    if a.pair:
        return pair_ab(a)
    N = a.n
    exps = [int(v) for v in a.exps.split(",")]
    for name in a.layers.split(","):
        C, h = LAYERS[name]
        x = torch.randn(N, h, h, C, device=DEV).to(BF)
        add = torch.randn(N, h, h, C, device=DEV).to(BF)
        w = (torch.randn(C, C, 3, 3, device=DEV) / (C * 9) ** 0.5).contiguous(memory_format=torch.channels_last)
        wf, wd = torch.empty(C * C * 9, device=DEV, dtype=BF), torch.empty(C * C * 9, device=DEV, dtype=BF)
        call("isic_conv_weight_prep_bf16", w, wf, wd, C, C, 3, 3)
        gf = 2.0 * N * h * h * C * 9 * C / 1e9
        modes = [("fwd+stats", wf, True, None), ("dgrad", wd, False, None)] + ([("dgrad+addend", wd, False, add)] if a.addend else [])
        if max(exps) >= 2:
            modes = [("dgrad", wd, False, None)]          # the timing ablations exist for the plain data gradient only
        for mode, wt, stats, ad in modes:
            outs, accs, fns, live = [], [], [], []
            for e in exps:
                out = torch.zeros(N, h, h, C, device=DEV, dtype=BF)
                acc = torch.zeros(2, 256, C, device=DEV, dtype=torch.float64)

                def run(e=e, out=out, acc=acc):
                    call("isic_test_conv2d_igemm_variant_bf16", x, wt, out, N, h, h, C, h, h, C, 3, 3, 1, 1, 1, ad,
                         acc[0] if stats else None, acc[1] if stats else None, 256 if stats else 0, e * 10000)
                try:
                    run()
                except IsicHipError as err:
                    print(f"{name} {mode:13s} exp {e}: {err}")
                    continue
                torch.cuda.synchronize()
                outs.append(out); accs.append(acc.sum(dim=1).clone()); fns.append(run); live.append(e)
            ts = interleaved(fns, a.iters)
            for e, out, st, t in zip(live, outs, accs, ts):
                same = "bit-equal" if torch.equal(out, outs[0]) and torch.equal(st, accs[0]) else \
                    f"DIFFERS max {float((out.float() - outs[0].float()).abs().max()):.3e}"
                print(f"{name} {mode:13s} exp {e}: {t:7.4f} ms  {gf / t:7.0f} TF/s  {same}", flush=True)


if __name__ == "__main__":
    main()
