"""HBM-roofline measurement of the patch-graph path at the BASELINE.json configs[3] shapes:
graphs of N = 196 nodes, D = 768 node features, k-NN k = 8, 3-layer GCN with F = 128, batched G graphs per launch.

Prints one JSON object: for the segmented-sum SpMM (forward and backward), the gated attention pool and the k-NN
build the achieved GB/s computed from the COMPULSORY bytes of SURVEY.md 8d (not the gather-expanded bytes) and the
fraction of the 8 TB/s HBM peak, plus graphs/s of the whole GraphMIL-gcn train step.  Developer / evidence tool:
    python tools/gnn_bench.py [--graphs 2048] [--iters 20]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-isic_amd"))
import numpy as np
import torch

HBM_PEAK_GBS = 8000.0


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3      # seconds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=2048, help="graphs per launch")
    ap.add_argument("--nodes", type=int, default=196)
    ap.add_argument("--feat", type=int, default=768)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--step-graphs", type=int, default=256, help="graphs per optimizer step of the train-step timing")
    ap.add_argument("--step-only", action="store_true", help="skip the kernel micro-benchmarks (for rocprofv3 of the train step)")
    a = ap.parse_args()
    from isic_hip import ops, optim
    from isic_hip.bags import BagOffsets
    from isic_hip.graph import GraphBatch, knn_indices, spmm
    from isic_hip.lib import call
    from gnn_models import GraphMIL

    dev = torch.device("cuda:0")
    G, N, D, F, k = a.graphs, a.nodes, a.feat, a.hidden, a.k
    gen = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(G * N, D, device=dev, generator=gen)
    offs = BagOffsets.from_lengths([N] * G, dev)
    res = {"graphs_per_launch": G, "nodes": N, "feat": D, "hidden": F, "k": k, "hbm_peak_GBs": HBM_PEAK_GBS}

    if a.step_only:
        G = a.step_graphs
        x = x[: G * N]
        offs = BagOffsets.from_lengths([N] * G, dev)
    # ---- k-NN build (03_build_graphs.py:37-54): compulsory bytes N*D*4 read + N*k*8 written per graph
    t = timeit(lambda: knn_indices(x, offs, k), 1 if a.step_only else a.iters)
    nn_idx = knn_indices(x, offs, k)                        # [G*N, k] node ids local to the graph
    nn_idx = nn_idx + (torch.arange(G * N, device=dev) // N * N).view(-1, 1)
    b = G * (N * D * 4 + N * k * 8)
    res["knn_build"] = {"ms": t * 1e3, "GBs": b / t / 1e9, "frac_hbm": b / t / 1e9 / HBM_PEAK_GBS,
                        "gflops": 2.0 * G * N * N * D / t / 1e9, "bytes_per_graph": b / G}
    src = torch.arange(G * N, device=dev).repeat_interleave(k)
    ei = torch.stack([src, nn_idx.reshape(-1).to(torch.int64)])
    graph = GraphBatch(ei, G * N)
    E = ei.shape[1] + G * N                                 # with self loops

    # ---- segmented-sum SpMM, forward and backward (GCNConv aggregation, 05_train_gnns.py:82,184-185)
    if a.step_only:
        a.iters = 1
    h = torch.randn(G * N, F, device=dev, generator=gen, requires_grad=True)
    comp = G * N * F * 4 * 2 + E * 4 * 2 + (G * N + 1) * 4  # read h + write out + col + val + rowptr
    t = timeit(lambda: spmm(h.detach(), graph), a.iters)
    res["spmm_fwd"] = {"ms": t * 1e3, "GBs": comp / t / 1e9, "frac_hbm": comp / t / 1e9 / HBM_PEAK_GBS,
                       "compulsory_bytes_per_graph": comp / G, "gather_expanded_bytes_per_graph": E * F * 4 / G}
    out = spmm(h, graph)
    dy = torch.randn_like(out)
    t = timeit(lambda: torch.autograd.grad(out, h, dy, retain_graph=True), a.iters)
    res["spmm_bwd"] = {"ms": t * 1e3, "GBs": comp / t / 1e9, "frac_hbm": comp / t / 1e9 / HBM_PEAK_GBS}

    # ---- gated attention pool (GraphMIL pooling, 05_train_gnns.py:198-213): 4 heads, A = 128
    heads, A, C = 4, 128, 7
    hh = torch.randn(G * N, F, device=dev, generator=gen)
    tt = torch.randn(G * N, heads * A, device=dev, generator=gen)     # tanh(h W2^T + b2), all heads
    w3 = torch.randn(heads, A, device=dev, generator=gen) * 0.1
    b3 = torch.zeros(heads, device=dev)
    att = torch.empty(G * N, heads, device=dev)
    z = torch.empty(G, heads, F, device=dev)
    comp_pool = G * N * (F + heads * A) * 4 + G * N * heads * 4 + G * heads * F * 4

    def pool():
        call("isic_attn_pool_fwd", hh, tt, w3, b3, None, None, offs.device, G, F, A, heads, 0, offs.max_bag, att, z, None,
             None, None, None)
    try:
        t = timeit(pool, a.iters)
        res["attn_pool_fwd"] = {"ms": t * 1e3, "GBs": comp_pool / t / 1e9, "frac_hbm": comp_pool / t / 1e9 / HBM_PEAK_GBS,
                                "bytes_per_graph": comp_pool / G}
    except Exception as e:                                  # signature drift must not hide the other numbers
        res["attn_pool_fwd"] = {"error": str(e)[:200]}

    # ---- whole train step: GraphMIL gcn, 3 layers, graphs resident in HBM with their CSR built once
    Gs = a.step_graphs
    model = GraphMIL(input_dim=D, gnn_type="gcn", gnn_hidden=F, gnn_layers=3, gnn_dropout=0.5, gnn_heads=4, att_dim=128,
                     att_heads=4, pool_dropout=0.2, classifier_dim=128, classifier_light=True, num_classes=7).to(dev)
    model.train()
    if hasattr(model, "set_dropout_state"):
        model.set_dropout_state(seed=1, step=0)
    opt = optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    xs = x[: Gs * N]
    offs_s = BagOffsets.from_lengths([N] * Gs, dev)
    keep = ei[0] < Gs * N
    graph_s = GraphBatch(ei[:, keep], Gs * N)
    y = (torch.arange(Gs, device=dev) % 7)

    def step():
        opt.zero_grad()
        probs, _ = model(xs, offsets=offs_s, graph=graph_s)
        loss = ops.cross_entropy_from_probs(probs, y)
        loss.backward()
        opt.step()
    t = timeit(step, 20 if a.step_only else max(3, a.iters // 2))
    res["graphmil_gcn3_train_step"] = {"graphs_per_step": Gs, "ms": t * 1e3, "graphs_per_s": Gs / t}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
