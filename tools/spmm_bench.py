"""The GCN aggregation launch (isic_spmm_csr_f32) alone, at the BASELINE.json configs[3] shapes: G graphs x 196 nodes, k-NN
k = 8 on 768-d synthetic embeddings (as bench.py draws them), F = 128.  Both orientations of the operator -- one has k + 1
entries in every row, the other's rows are k-NN in-degrees (hubs) -- timed as `--iters` back-to-back launches through the
C ABI between one pair of events; GB/s over the COMPULSORY bytes of SURVEY.md 8d.  Developer / evidence tool:
    python tools/spmm_bench.py [--graphs 256] [--iters 50]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-isic_amd"))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=256)
    ap.add_argument("--nodes", type=int, default=196)
    ap.add_argument("--feat", type=int, default=768)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    from isic_hip.bags import BagOffsets
    from isic_hip.graph import GraphBatch, knn_indices
    from isic_hip.lib import call
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    G, N, D, F, k = a.graphs, a.nodes, a.feat, a.hidden, a.k
    x = torch.randn(G * N, D, device=dev, generator=gen)
    offs = BagOffsets.from_lengths([N] * G, dev)
    nn_idx = knn_indices(x, offs, k) + (torch.arange(G * N, device=dev) // N * N).view(-1, 1)
    src = torch.arange(G * N, device=dev).repeat_interleave(k)
    graph = GraphBatch(torch.stack([src, nn_idx.reshape(-1).to(torch.int64)]), G * N)
    n = G * N
    E = src.numel() + n
    h = torch.randn(n, F, device=dev, generator=gen)
    out = torch.empty_like(h)
    comp = n * F * 4 * 2 + E * 4 * 2 + (n + 1) * 4
    print(f"{G} graphs x {N} nodes, k = {k}, F = {F}: {E} stored entries, compulsory {comp / 1e6:.1f} MB per launch")
    for name, (rp, c, v) in (("forward CSR", (graph.rowptr, graph.col, graph.val)),
                             ("transposed CSR", (graph.rowptr_t, graph.col_t, graph.val_t))):
        deg = (rp[1:] - rp[:-1]).float()
        fn = lambda: call("isic_spmm_csr_f32", rp, c, v, h, None, out, n, F, 1.0, None, 0.0)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / a.iters * 1e-3
        print(f"{name:15s} row length {int(deg.min())}..{int(deg.max())} (mean {deg.mean().item():.1f}, "
              f"{(deg > 16).float().mean().item() * 100:.1f} % > 16): {t * 1e6:6.1f} us = {comp / t / 1e12:5.2f} TB/s "
              f"({comp / t / 8e12:.2f} of 8 TB/s)")


if __name__ == "__main__":
    main()
