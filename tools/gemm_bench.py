"""A/B timing of the fp32 GEMM kernels on the graph path's products (256 graphs x 196 nodes per step; developer tool):
the 64 x 64 x 16 register-staged kernel (gemm_f32.hip), the persistent 256 x 128 x 32 LDS-DMA kernel (gemm_f32p.hip), the
register-fed A^T B split-K kernel (gemm_f32t.hip), the row-panel kernel for K <= 128 (gemm_f32r.hip) and the shipped choice;
every result is also compared with an fp64 product.

    python tools/gemm_bench.py [--iters 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-isic_amd"))
import torch
from isic_hip.lib import IsicHipError, call

DEV = "cuda:0"
PEAK = 157.3      # TFLOP/s, v_mfma_f32_16x16x4_f32, MI355X_MICROARCH.md

SHAPES = [  # (name, M, N, K, transA, transB, bias)
    ("input_proj fwd  x W^T", 50176, 128, 768, 0, 1, 1),
    ("gcn lin fwd     h W^T", 50176, 128, 128, 0, 1, 0),
    ("att heads fwd   h W^T", 50176, 512, 128, 0, 1, 1),
    ("att heads dX    dY W", 50176, 128, 512, 0, 0, 0),
    ("gcn lin dX      dY W", 50176, 128, 128, 0, 0, 0),
    ("input_proj dW   dY^T X", 128, 768, 50176, 1, 0, 0),
    ("gcn lin dW      dY^T X", 128, 128, 50176, 1, 0, 0),
    ("att heads dW    dY^T X", 512, 128, 50176, 1, 0, 0),
    ("mil head fwd", 2048, 128, 512, 0, 1, 1),
    ("(scaling) 4 x rows", 200704, 128, 128, 0, 1, 0),
    ("(scaling) rows / 4", 12544, 128, 128, 0, 1, 0),
    ("(scaling) dW 4 x K", 128, 128, 200704, 1, 0, 0),
]


VARIANTS = [(1, "64x64x16"), (2, "persistent"), (3, "A^T B regs"), (4, "row-panel"), (0, "shipped")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    ws = torch.empty(256 << 20, device=DEV, dtype=torch.uint8)
    print(f"{'product':24s} {'M':>6s} {'N':>5s} {'K':>6s} " + " ".join(f"{n:>17s}" for _, n in VARIANTS) + "   (us, fraction of 157.3 TFLOP/s; max |err| vs fp64)")
    for name, M, N, K, ta, tb, hb in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device=DEV)
        B = torch.randn((N, K) if tb else (K, N), device=DEV)
        C = torch.empty(M, N, device=DEV)
        bias = torch.randn(N, device=DEV) if hb else None
        ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
        if hb:
            ref = ref + bias.double()
        fl = 2.0 * M * N * K
        cells = []
        for variant, _ in VARIANTS:
            def run():
                call("isic_test_gemm_f32_variant", variant, ta, tb, M, N, K, A, A.shape[1], B, B.shape[1], C, N, bias, 0, 0.0,
                     ws, ws.numel())
            try:
                C.fill_(float("nan"))
                for _ in range(3):
                    run()
            except IsicHipError:
                cells.append(f"{'--':>17s}")
                continue
            err = (C.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            cells.append(f"{us:7.1f} {fl / us / 1e6 / PEAK:4.2f} {err:4.0e}")
        print(f"{name:24s} {M:6d} {N:5d} {K:6d} " + " ".join(cells))


if __name__ == "__main__":
    main()
