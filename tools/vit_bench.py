"""ViT-S/16 fp16 encoder throughput at the BASELINE.json configs[4] shape (images of 224x224 -> 196 x 384 tokens):
images/s, algorithmic TFLOP/s of the whole forward, and the per-kernel times of one block.  Developer tool:
    python tools/vit_bench.py [--n 2048] [--iters 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-isic_amd"))
import torch
from isic_hip.lib import call
from isic_hip.vit import ViTSmallEncoder

DEV, F16 = "cuda:0", torch.float16


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
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    x = torch.randn(a.n, 3, 224, 224, device=DEV)
    for fold in (True, "stats", False):      # LayerNorms inside the products / statistics-only passes / passes of their own
        enc = ViTSmallEncoder(fold_layernorm=fold).to(DEV)
        ms = timeit(lambda: enc.run_tokens(x), a.iters)
        fl = enc.flops_per_image() * a.n
        print(f"ViT-S/16 fp16 forward, fold_layernorm={fold!s:5}: {a.n} images in {ms:.2f} ms = {a.n / ms * 1e3:.0f} images/s, "
              f"{fl / ms / 1e9:.0f} TFLOP/s algorithmic ({fl / ms / 1e9 / 2500:.2f} of 2.5 PF dense fp16)")
        del enc
    M, D = a.n * 196, 384
    h = torch.randn(M, D, device=DEV).to(F16)
    big = torch.randn(M, 4 * D, device=DEV).to(F16)
    qkv = torch.randn(M, 3 * D, device=DEV).to(F16)
    out = torch.empty(M, D, device=DEV, dtype=F16)
    gam, bet = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    for name, N, K, act, res, src, dst in (("qkv 384->1152", 3 * D, D, 0, None, h, qkv), ("proj 384->384 + residual", D, D, 0, h, h, out),
                                           ("fc1 384->1536 + GELU", 4 * D, D, 1, None, h, big), ("fc2 1536->384 + residual", D, 4 * D, 0, h, big, out)):
        W = (torch.randn(N, K, device=DEV) * 0.02).to(F16)
        b = torch.zeros(N, device=DEV)
        t = timeit(lambda: call("isic_gemm_f16", src, W, b, res, dst, M, N, K, act, 0), a.iters)
        print(f"  gemm {name:28s} {t:7.3f} ms  {2.0 * M * N * K / t / 1e9:6.0f} TFLOP/s")
    # the LayerNorm-folded forms of the same four products
    st = torch.empty(M, 6, 2, device=DEV)
    call("isic_gemm_f16_stats", h, (torch.randn(D, D, device=DEV) * 0.02).to(F16), torch.zeros(D, device=DEV), h, out, st, M, D, D, 0, 0)
    for name, N, K, act, src, dst in (("qkv  LN folded", 3 * D, D, 0, out, qkv), ("fc1  LN folded + GELU", 4 * D, D, 1, out, big)):
        W = (torch.randn(N, K, device=DEV) * 0.02).to(F16)
        b, c = torch.zeros(N, device=DEV), W.float().sum(1)
        t = timeit(lambda: call("isic_gemm_f16_ln", src, W, b, c, st, 6, dst, M, N, K, act, 1e-6), a.iters)
        print(f"  gemm {name:28s} {t:7.3f} ms  {2.0 * M * N * K / t / 1e9:6.0f} TFLOP/s")
    for name, N, K, src in (("proj + residual + row sums", D, D, h), ("fc2  + residual + row sums", D, 4 * D, big)):
        W = (torch.randn(N, K, device=DEV) * 0.02).to(F16)
        b = torch.zeros(N, device=DEV)
        t = timeit(lambda: call("isic_gemm_f16_stats", src, W, b, h, out, st, M, N, K, 0, 0), a.iters)
        print(f"  gemm {name:28s} {t:7.3f} ms  {2.0 * M * N * K / t / 1e9:6.0f} TFLOP/s")
    t = timeit(lambda: call("isic_row_stats_f16", h, st, M, D, 1e-6), a.iters)
    print(f"  row statistics 384                  {t:7.3f} ms  {2.0 * M * D / t / 1e6:6.0f} GB/s")
    t = timeit(lambda: call("isic_attention_f16", qkv, out, a.n, 196, 6, 64), a.iters)
    print(f"  attention 6 heads x 196 tokens      {t:7.3f} ms  {4.0 * a.n * 6 * 196 * 196 * 64 / t / 1e9:6.0f} TFLOP/s")
    t = timeit(lambda: call("isic_layernorm_f16", h, gam, bet, out, None, M, D, 1e-6), a.iters)
    print(f"  layernorm 384                       {t:7.3f} ms  {2.0 * M * D * 2 / t / 1e6:6.0f} GB/s")


if __name__ == "__main__":
    main()
