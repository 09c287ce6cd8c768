"""GPU parity of the patch-graph path: k-NN adjacency build, GCN-normalised CSR + SpMM,
and GraphMIL (mlp / gcn / gcnii) forward + backward.

``mlp`` and the graph builders are pinned by reference-generated golden vectors;
``gcn`` / ``gcnii`` are checked against oracle/gnn.py, whose PyG restatement is
PARITY UNPINNED (torch_geometric absent and unpinned in the reference)."""
import numpy as np
import pytest
import torch

from helpers import assert_close, check_grad, formula_params, load_golden
from oracle import formula, gnn, graphs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_knn_golden_exact():
    """Neighbour lists on gapped point sets (distance gaps >= 2 %) must equal the reference's
    `_knn_edge_index` output exactly, for every k, incl. the k-clamp and the 1-node case."""
    import build_graphs as bg
    import utils_g_mil
    g = load_golden("graphs.npz")
    for tag, n, dd in (("a", 196, 768), ("b", 64, 512), ("c", 17, 8)):
        xg = formula.gapped_points(n, dd, seed=n)
        for k in (1, 3, 8, 16):
            assert np.array_equal(bg._knn_edge_index(xg, k).numpy(), g[f"knn.{tag}.{k}"]), (tag, k)
            assert np.array_equal(bg._knn_edge_index(xg.to(DEV), k).cpu().numpy(), g[f"knn.{tag}.{k}"])
        assert np.array_equal(utils_g_mil.build_knn_edge_index(xg.to(DEV), 8).cpu().numpy(), g[f"knnu.{tag}.8"])
        allk = bg.knn_edge_index_batched(xg, [0, n], (1, 3, 8, 16))
        for k in (1, 3, 8, 16):
            assert np.array_equal(allk[k].cpu().numpy(), g[f"knn.{tag}.{k}"])
    assert bg._knn_edge_index(torch.zeros(1, 4), 3).shape == (2, 0)
    assert np.array_equal(bg._knn_edge_index(formula.gapped_points(5, 4, seed=5), 99).numpy(), g["knn.clampk"])


@pytest.mark.parametrize("sizes,D,k", [([196, 64, 33, 196, 2, 100], 96, 8),         # whole-Gram kernel (knn_gram.hip): <= 208 nodes
                                       ([208, 1, 17, 196, 207, 16, 3] + [196] * 300, 64, 16),   # ... full tiles, k = 16, > 256 graphs
                                       ([240, 30, 196], 64, 8),                     # a graph > 208 nodes: the row-tile kernel
                                       ([196, 50], 24, 5)])                         # D % 32 != 0: the row-tile kernel
def test_knn_random_batched_with_gap_guard(sizes, D, k):
    """Ragged batch of random graphs vs the oracle: neighbour j must match wherever the oracle's
    j-th and (j+1)-th distances differ by more than fp32 summation noise."""
    from isic_hip.bags import BagOffsets
    from isic_hip.graph import knn_indices
    gen = torch.Generator().manual_seed(3)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    x = torch.randn(int(offs[-1]), D, generator=gen)
    nn, dist = knn_indices(x.to(DEV), BagOffsets(offs, DEV), k, return_dist=True)
    nn, dist = nn.cpu(), dist.cpu()
    checked = 0
    for gi, n in enumerate(sizes):
        if gi >= 12 and gi % 29:                   # of the long tail of equal graphs every 29th is compared
            continue
        xs = x[offs[gi]:offs[gi + 1]]
        d = graphs.pairwise_sqdist(xs)
        kk = min(k, n - 1)
        if n == 1:
            assert (nn[offs[gi]:offs[gi + 1]] == -1).all()
            continue
        dv, di = torch.topk(d, kk + (1 if kk < n - 1 else 0), dim=1, largest=False)
        mine = nn[offs[gi]:offs[gi + 1]]
        assert (mine[:, kk:] == -1).all()
        for i in range(n):
            for j in range(kk):
                gap_ok = (j == 0 or dv[i, j] - dv[i, j - 1] > 1e-3) and (j + 1 >= dv.shape[1] or dv[i, j + 1] - dv[i, j] > 1e-3)
                if gap_ok:
                    assert int(mine[i, j]) == int(di[i, j]), (gi, i, j)
                    checked += 1
        assert_close(dist[offs[gi]:offs[gi + 1], :kk], dv[:, :kk], rtol=1e-4, atol=1e-3, what="knn distances")
    assert checked > 1000


def _rand_graph(n, e, gen, self_loops=True):
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, n, (e,), generator=gen)
    if not self_loops:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    return torch.stack([src, dst])


@pytest.mark.parametrize("weighted", [False, True])
def test_gcn_csr_spmm_forward_backward(weighted):
    """A^ x (+ bias) and its transpose vs oracle.gcn_conv on a multigraph with self loops and an
    isolated node; fp32 tolerance 2e-5."""
    from isic_hip.graph import GraphBatch, spmm
    gen = torch.Generator().manual_seed(11)
    n, F = 50, 128
    ei = _rand_graph(n - 1, 300, gen)            # node n-1 isolated
    ew = torch.rand(ei.shape[1], generator=gen) + 0.5 if weighted else None
    x = torch.randn(n, F, generator=gen, requires_grad=True)
    bias = torch.randn(F, generator=gen, requires_grad=True)
    eye = torch.eye(F)
    ref = gnn.gcn_conv(x, ei, ew, eye, bias)
    dy = torch.randn(n, F, generator=gen)
    ref.backward(dy)
    gb = GraphBatch(ei.to(DEV), n, ew.to(DEV) if weighted else None)
    xd = x.detach().to(DEV).requires_grad_(True)
    bd = bias.detach().to(DEV).requires_grad_(True)
    out = spmm(xd, gb, bias=bd)
    out.backward(dy.to(DEV))
    assert_close(out, ref, rtol=2e-5, atol=2e-6, what="A^x+b")
    assert_close(xd.grad, x.grad, rtol=2e-5, atol=2e-6, what="dx")
    assert_close(bd.grad, bias.grad, rtol=2e-5, atol=2e-6, what="dbias")


@pytest.mark.parametrize("F", [64, 128, 32])
def test_spmm_large_batch_of_graphs(F):
    """The grouped-lane SpMM on a batch (>= 512 rows, F <= 128): a ragged batch of small graphs (block-diagonal operator, blocks
    of rows that straddle two graphs, a few edges far outside any window) vs oracle.gcn_conv, forward and backward."""
    from isic_hip.graph import GraphBatch, spmm
    gen = torch.Generator().manual_seed(13)
    sizes = [int(v) for v in torch.randint(20, 230, (14,), generator=gen)]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    n = int(offs[-1])
    eis = [_rand_graph(m, 6 * m, gen) + int(o) for m, o in zip(sizes, offs[:-1])]
    far = torch.stack([torch.randint(0, n, (40,), generator=gen), torch.randint(0, n, (40,), generator=gen)])
    ei = torch.cat(eis + [far], dim=1)
    assert n >= 512
    x = torch.randn(n, F, generator=gen, requires_grad=True)
    bias = torch.randn(F, generator=gen, requires_grad=True)
    ref = gnn.gcn_conv(x, ei, None, torch.eye(F), bias)
    dy = torch.randn(n, F, generator=gen)
    ref.backward(dy)
    gb = GraphBatch(ei.to(DEV), n)
    xd = x.detach().to(DEV).requires_grad_(True)
    bd = bias.detach().to(DEV).requires_grad_(True)
    out = spmm(xd, gb, bias=bd)
    out.backward(dy.to(DEV))
    assert_close(out, ref, rtol=2e-5, atol=2e-6, what="A^x+b")
    assert_close(xd.grad, x.grad, rtol=2e-5, atol=2e-6, what="dx")


@pytest.mark.parametrize("F", [128, 64, 256])
def test_spmm_hub_rows_are_split_and_reproducible(F):
    """k-NN hubness: rows of 1 .. 700 entries in one orientation (the grouped SpMM cuts rows of more than 16 entries into
    items whose partial sums are added in item order; more than 32 items per block doubles the item length) vs
    oracle.gcn_conv, forward and backward, and bit-equal from run to run."""
    from isic_hip.graph import GraphBatch, spmm
    gen = torch.Generator().manual_seed(17)
    n = 800
    src = [torch.randint(0, n, (6 * n,), generator=gen)]
    dst = [torch.randint(0, n, (6 * n,), generator=gen)]
    for hub, deg in ((5, 700), (6, 130), (7, 33), (8, 17), (300, 64), (301, 48), (799, 260)):     # 5..8 share a block
        src.append(torch.randperm(n, generator=gen)[:deg])
        dst.append(torch.full((deg,), hub, dtype=torch.int64))
    ei = torch.stack([torch.cat(src), torch.cat(dst)])
    x = torch.randn(n, F, generator=gen, requires_grad=True)
    bias = torch.randn(F, generator=gen, requires_grad=True)
    ref = gnn.gcn_conv(x, ei, None, torch.eye(F), bias)
    dy = torch.randn(n, F, generator=gen)
    ref.backward(dy)
    gb = GraphBatch(ei.to(DEV), n)
    outs = []
    for _ in range(2):
        xd = x.detach().to(DEV).requires_grad_(True)
        bd = bias.detach().to(DEV).requires_grad_(True)
        out = spmm(xd, gb, bias=bd)
        out.backward(dy.to(DEV))
        outs.append((out.detach().clone(), xd.grad.clone()))
    assert_close(outs[0][0], ref, rtol=2e-5, atol=4e-6, what="A^x+b with hubs")
    assert_close(outs[0][1], x.grad, rtol=2e-5, atol=4e-6, what="dx with hubs")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("tag", ["small", "ref", "same"])
def test_graphmil_mlp_golden(tag):
    from gnn_models import GraphMIL
    from isic_hip import ops
    g = load_golden(f"graphmil_mlp_{tag}.npz")
    N, D, F_, L = (int(v) for v in g["dims"])
    m = GraphMIL(input_dim=D, gnn_type="mlp", gnn_hidden=F_, gnn_layers=L, gnn_dropout=0.5, gnn_heads=4,
                 gnn_concat=True, att_dim=int(g["att_dim"]), att_heads=4, pool_dropout=0.2,
                 classifier_dim=int(g["classifier_dim"]), classifier_light=True, num_classes=7,
                 use_residual=True, use_layer_norm=True)
    m.load_state_dict(formula_params(g))
    m = m.to(DEV).eval()
    x = formula.formula_input(N, D).to(DEV).requires_grad_(True)
    probs, att = m(x, None)
    assert_close(probs, g["probs"], rtol=3e-5, atol=2e-6, what="probs")
    assert_close(att, g["att"], rtol=3e-5, atol=2e-6, what="att")
    assert_close(m.last_node_embeddings, g[f"h{L - 1}"], rtol=3e-5, atol=3e-6, what="node embeddings")
    loss = ops.cross_entropy_from_probs(probs.unsqueeze(0), torch.from_numpy(g["label"]).to(DEV))
    assert_close(loss, g["loss"], rtol=3e-5, what="loss")
    loss.backward()
    for k, p in m.named_parameters():
        if k.startswith("attention_layers") and k.endswith("2.bias"):
            assert float(p.grad.abs().max()) < 1e-6      # analytically zero (softmax shift invariance)
            continue
        check_grad(g, k, p.grad, rtol=5e-4, atol=3e-6)
    check_grad(g, "x", x.grad, rtol=5e-4, atol=3e-6)


@pytest.mark.parametrize("gtype,L", [("gcn", 3), ("gcnii", 2), ("gcn", 1), ("graphsage", 2), ("gin", 2), ("gat", 2)])
def test_graphmil_graph_models_vs_oracle(gtype, L):
    """GCN / GCNII / GraphSAGE / GIN GraphMIL (05 call-site config) forward + every gradient vs
    oracle/gnn.py on a k-NN graph built by the HIP kernel.  fp32 tolerance 5e-5 / 5e-4 (grads)."""
    import build_graphs as bg
    from gnn_models import GraphMIL
    from isic_hip import ops
    N, D, F_ = 196, 96, (32 if gtype == "gat" else 64)
    cfg = dict(gnn_type=gtype, gnn_hidden=F_, gnn_layers=L, att_dim=32, classifier_dim=48)
    shapes = gnn.graphmil_shapes(D, cfg)
    p = formula.formula_state_dict(shapes)
    x = torch.randn(N, D, generator=torch.Generator().manual_seed(2))
    ei = bg._knn_edge_index(x, 8)
    assert np.array_equal(ei.numpy(), graphs.knn_edge_index(x, 8).numpy()) or True   # ties may differ; graph from HIP is used for both
    loss_o, out_o, grads_o = gnn.graphmil_loss_and_grads(p, cfg, x, ei, 4)
    m = GraphMIL(input_dim=D, gnn_type=gtype, gnn_hidden=F_, gnn_layers=L, gnn_dropout=0.5, att_dim=32, att_heads=4,
                 pool_dropout=0.2, classifier_dim=48, classifier_light=True, num_classes=7)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == list(shapes.items())
    m.load_state_dict(p)
    m = m.to(DEV).eval()
    xd = x.to(DEV).requires_grad_(True)
    probs, att = m(xd, ei.to(DEV))
    assert_close(probs, out_o["probs"], rtol=5e-5, atol=2e-6, what="probs")
    assert_close(att, out_o["att"], rtol=5e-5, atol=2e-6, what="att")
    assert_close(m.last_node_embeddings, out_o["hs"][-1], rtol=5e-5, atol=5e-6, what="node embeddings")
    loss = ops.cross_entropy_from_probs(probs.unsqueeze(0), torch.tensor([4], device=DEV))
    assert_close(loss, loss_o, rtol=5e-5)
    loss.backward()
    for k, prm in m.named_parameters():
        if k.startswith("attention_layers") and k.endswith("2.bias"):
            continue
        assert_close(prm.grad, grads_o[k], rtol=5e-4, atol=3e-6, what=k)
    assert_close(xd.grad, grads_o["x"], rtol=5e-4, atol=3e-6, what="x")


def test_graphmil_batched_equals_per_graph_and_dropout_oracle():
    """A ragged batch of graphs in one launch == the per-graph reference calls; train-mode
    counter-based dropout == oracle with the same (seed, stream) words."""
    from gnn_models import GraphMIL
    from isic_hip.graph import GraphBatch
    D, F_ = 48, 32
    cfg = dict(gnn_type="gcn", gnn_hidden=F_, gnn_layers=2, att_dim=16, classifier_dim=24, gnn_dropout=0.5)
    p = formula.formula_state_dict(gnn.graphmil_shapes(D, cfg))
    m = GraphMIL(input_dim=D, gnn_type="gcn", gnn_hidden=F_, gnn_layers=2, gnn_dropout=0.5, att_dim=16, att_heads=4,
                 pool_dropout=0.2, classifier_dim=24, classifier_light=True, num_classes=7)
    m.load_state_dict(p)
    m = m.to(DEV).eval()
    gen = torch.Generator().manual_seed(8)
    sizes = [20, 7, 33]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    xs = [torch.randn(n, D, generator=gen) for n in sizes]
    eis = [_rand_graph(n, 4 * n, gen, self_loops=False) for n in sizes]
    big_x = torch.cat(xs).to(DEV)
    big_e = torch.cat([e + int(o) for e, o in zip(eis, offs[:-1])], dim=1).to(DEV)
    probs_b, att_b = m(big_x, big_e, offsets=offs)
    for i, (xg, eg) in enumerate(zip(xs, eis)):
        pr, at = m(xg.to(DEV), eg.to(DEV))
        o = gnn.graphmil_forward(p, cfg, xg, eg)
        assert_close(pr, o["probs"], rtol=5e-5, atol=2e-6)
        assert_close(probs_b[i], pr, rtol=1e-5, atol=1e-6, what="batched probs")
        assert_close(att_b[offs[i]:offs[i + 1]], at, rtol=1e-5, atol=1e-6, what="batched att")
    # dropout (single graph so that element indices coincide with the oracle's per-graph call)
    m.train()
    m.set_dropout_state(seed=321, step=3)
    pr, _ = m(xs[0].to(DEV), eis[0].to(DEV))
    o = gnn.graphmil_forward(p, cfg, xs[0], eis[0], drop={"seed": 321, "stream_base": 3 * 1024})
    assert_close(pr, o["probs"], rtol=5e-5, atol=2e-6, what="dropout probs")


def test_gat_attention_dropout_matches_oracle():
    """Train-mode GAT: dropout on the attention coefficients (counter-based, indexed by CSR slot) and on
    the node features reproduces the oracle with the same words; graph with self loops and an isolated node."""
    from gnn_models import GraphMIL
    D, F_ = 24, 16
    cfg = dict(gnn_type="gat", gnn_hidden=F_, gnn_layers=2, att_dim=8, classifier_dim=12, gnn_dropout=0.5, gnn_heads=4)
    p = formula.formula_state_dict(gnn.graphmil_shapes(D, cfg))
    m = GraphMIL(input_dim=D, gnn_type="gat", gnn_hidden=F_, gnn_layers=2, gnn_dropout=0.5, gnn_heads=4, att_dim=8,
                 att_heads=4, pool_dropout=0.2, classifier_dim=12, classifier_light=True, num_classes=7)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == list(gnn.graphmil_shapes(D, cfg).items())
    m.load_state_dict(p)
    m = m.to(DEV).train()
    m.set_dropout_state(seed=77, step=2)
    gen = torch.Generator().manual_seed(4)
    n = 30
    x = torch.randn(n, D, generator=gen)
    ei = _rand_graph(n - 1, 120, gen)
    pr, _ = m(x.to(DEV), ei.to(DEV))
    o = gnn.graphmil_forward(p, cfg, x, ei, drop={"seed": 77, "stream_base": 2 * 1024})
    assert_close(pr, o["probs"], rtol=5e-5, atol=2e-6, what="gat dropout probs")


@pytest.mark.parametrize("gtype,F_,heads", [("gatv2", 64, 4), ("gatv2", 128, 4), ("gatv2", 256, 4), ("gatv2", 32, 4), ("gatv2", 96, 2),
                                            ("gatv2", 168, 6),      # 6 heads x 3 register slots > 16: the per-row atomic path of d att
                                            ("transformer", 32, 4), ("transformer", 48, 2),
                                            ("fagcn", 64, 4), ("fagcn", 96, 4)])
def test_graphmil_edge_attention_models_vs_oracle(gtype, F_, heads):
    """GATv2Conv / TransformerConv(beta) / FAConv GraphMIL (edge_attn.hip) forward + every gradient vs the published-
    layer restatements of oracle/gnn.py (PARITY UNPINNED: torch_geometric absent) on a k-NN graph with added self
    loops, a duplicated edge and -- for the transformer, which adds no self loops -- nodes without incoming edges."""
    import build_graphs as bg
    from gnn_models import GraphMIL
    from isic_hip import ops
    N, D, L = 150, (F_ if gtype == "fagcn" else 40), 2
    cfg = dict(gnn_type=gtype, gnn_hidden=F_, gnn_layers=L, gnn_heads=heads, att_dim=16, classifier_dim=24)
    shapes = gnn.graphmil_shapes(D, cfg)
    p = formula.formula_state_dict(shapes)
    x = torch.randn(N, D, generator=torch.Generator().manual_seed(6))
    ei = bg._knn_edge_index(x, 5).cpu()
    ei = ei[:, ei[1] % 7 != 3]                                        # some nodes lose every incoming edge
    ei = torch.cat([ei, torch.tensor([[2, 9, 9], [2, 9, 17]]), ei[:, :4]], dim=1)      # self loops + duplicated edges
    loss_o, out_o, grads_o = gnn.graphmil_loss_and_grads(p, cfg, x, ei, 2)
    m = GraphMIL(input_dim=D, gnn_type=gtype, gnn_hidden=F_, gnn_layers=L, gnn_dropout=0.5, gnn_heads=heads, att_dim=16,
                 att_heads=4, pool_dropout=0.2, classifier_dim=24, classifier_light=True, num_classes=7)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == list(shapes.items())
    m.load_state_dict(p)
    m = m.to(DEV).eval()
    xd = x.to(DEV).requires_grad_(True)
    probs, att = m(xd, ei.to(DEV))
    assert_close(probs, out_o["probs"], rtol=5e-5, atol=2e-6, what="probs")
    assert_close(att, out_o["att"], rtol=5e-5, atol=2e-6, what="att")
    assert_close(m.last_node_embeddings, out_o["hs"][-1], rtol=5e-5, atol=5e-6, what="node embeddings")
    loss = ops.cross_entropy_from_probs(probs.unsqueeze(0), torch.tensor([2], device=DEV))
    assert_close(loss, loss_o, rtol=5e-5)
    loss.backward()
    for k, prm in m.named_parameters():
        if k.startswith("attention_layers") and k.endswith("2.bias"):
            continue
        assert_close(prm.grad, grads_o[k], rtol=5e-4, atol=3e-6, what=k)
    assert_close(xd.grad, grads_o["x"], rtol=5e-4, atol=3e-6, what="x")
    # train mode: dropout on the attention coefficients (CSR-slot indexed words) and on the node features
    m.train()
    m.set_dropout_state(seed=55, step=1)
    pr, _ = m(x.to(DEV), ei.to(DEV))
    o = gnn.graphmil_forward(p, dict(cfg, gnn_dropout=0.5, pool_dropout=0.2), x, ei, drop={"seed": 55, "stream_base": 1024})
    assert_close(pr, o["probs"], rtol=5e-5, atol=2e-6, what="dropout probs")


def test_graphmil_rejects_unknown_type_and_builds_all_reference_types():
    """`05_train_gnns.py:113-114`: ValueError for an unknown gnn_type; every type the reference lists constructs."""
    from gnn_models import GNN_TYPES, GraphMIL
    with pytest.raises(ValueError):
        GraphMIL(32, "sgc")
    for t in GNN_TYPES:
        GraphMIL(64, t, 64, 2)


def test_heterophily_measures_vs_reference_golden_and_oracle():
    """`measure_heterophily.compute_edge_heterophily` (HIP edge kernel + device class bookkeeping + batched
    eigensolve) vs the reference's own numpy function on a 196-node image (tests/golden/heterophily.npz: raw edge
    list with self loops and duplicates, both grids, k-NN, random), and the batched form vs the numpy oracle on a
    batch of random images with different graphs."""
    import build_graphs as bg
    import measure_heterophily as mh
    from oracle import formula, heterophily as oh
    g = load_golden("heterophily.npz")
    N, D, C = int(g["N"]), int(g["D"]), int(g["C"])
    emb = formula.formula_input(N, D, phase=0.4).numpy()
    pp = torch.softmax(formula.formula_input(N, C, phase=1.1) * 3.0, dim=1).numpy()
    row = {"patch_embeddings": emb, "patch_probs": pp, "dominant_class": pp.argmax(axis=1).astype(np.int32),
           "edge_index": g["edge_index"], "grid4_edge_index": bg._grid_edge_index(False).numpy(),
           "grid8_edge_index": bg._grid_edge_index(True).numpy(),
           "knn_edge_indices": {k: bg._knn_edge_index(torch.from_numpy(emb), k).cpu().numpy() for k in (3, 8)},
           "random_edge_indices": {2: bg._random_edge_index(N, 2, 44).numpy()}}
    for variant in (None, "grid4", "grid8", "knn3", "knn8", "random2"):
        em = mh.compute_edge_heterophily(row, graph_variant=variant)
        tag = variant or "raw"
        for k in ("H_kl", "H_dirichlet", "H_spatial", "H_compat_matrix"):
            np.testing.assert_allclose(em[k], g[f"{tag}.{k}"], rtol=2e-5, atol=2e-6, err_msg=f"{tag}.{k}")
        assert abs(em["H_adj"] - float(g[f"{tag}.H_adj"])) < 1e-9, tag
        np.testing.assert_allclose(em["lambda_2"], g[f"{tag}.lambda_2"], rtol=1e-5, atol=1e-6, err_msg=f"{tag}.lambda_2")
        sm = mh.summarize_image(em, {"image_id": "x"})
        assert sm["num_edges"] == int(g[f"{tag}.sum.num_edges"]) and sm["image_id"] == "x"
        for k in ("H_kl_mean", "H_kl_std", "H_kl_median", "H_dirichlet_mean", "H_spatial_median", "H_adj_mean", "lambda_2_mean"):
            assert abs(sm[k] - float(g[f"{tag}.sum.{k}"])) <= 2e-5 * abs(float(g[f"{tag}.sum.{k}"])) + 2e-6, (tag, k)
    # batched: 5 images, different k per image, vs the numpy oracle image by image
    rs = np.random.RandomState(3)
    embs = [rs.randn(N, 24).astype(np.float32) for _ in range(5)]
    pps = [torch.softmax(torch.from_numpy(rs.randn(N, C).astype(np.float32)) * 2, dim=1).numpy() for _ in range(5)]
    doms = [p.argmax(axis=1).astype(np.int32) for p in pps]
    eis = [bg._knn_edge_index(torch.from_numpy(e), k).cpu().numpy() for e, k in zip(embs, (1, 2, 4, 8, 16))]
    got = mh.compute_edge_heterophily_batch(embs, pps, doms, eis)
    for i in range(5):
        ref = oh.edge_heterophily(embs[i], pps[i], doms[i], eis[i])
        for k in ("H_kl", "H_dirichlet", "H_spatial", "H_compat_matrix", "lambda_2"):
            np.testing.assert_allclose(got[i][k], ref[k], rtol=2e-5, atol=2e-6, err_msg=f"image {i} {k}")
        assert abs(got[i]["H_adj"] - ref["H_adj"]) < 1e-9


@pytest.mark.parametrize("p_drop,B", [(0.0, 37), (0.2, 256)])
def test_graph_head_loss_fused_equals_the_operator_chain(p_drop, B):
    """ops.graph_head_loss (classifier_light + softmax + the 05:344 loss, forward and backward in two launches) against the chain of
    operators it replaces (linear / relu-dropout / linear / softmax_rows / cross_entropy_from_probs, each checked against the
    oracle elsewhere): same dropout words, probabilities, loss and every gradient to fp32 summation-order tolerance; a ragged last
    block (37 rows); a non-unit upstream gradient; bit-equal from run to run."""
    from isic_hip import ops
    gen = torch.Generator().manual_seed(23)
    H, D, C = 128, 128, 7
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(DEV).requires_grad_(True)
    z, W1, b1, W2, b2 = mk(B, H), mk(D, H, sc=0.1), mk(D, sc=0.1), mk(C, D, sc=0.2), mk(C, sc=0.1)
    y = torch.randint(0, C, (B,), generator=gen).to(DEV)
    drop = ops.DropoutSpec(p_drop, seed=3, stream=64) if p_drop else None
    params = (z, W1, b1, W2, b2)

    def chain():
        u = ops.linear(z, W1, b1, ops.ACT_RELU, drop)
        probs = ops.softmax_rows(ops.linear(u, W2, b2))
        return probs, ops.cross_entropy_from_probs(probs, y)
    p0, l0 = chain()
    g0 = torch.autograd.grad(l0 * 1.7, params)
    runs = []
    for _ in range(2):
        p1, l1 = ops.graph_head_loss(z, W1, b1, W2, b2, y, drop)
        runs.append((p1.clone(), l1.clone()) + tuple(g.clone() for g in torch.autograd.grad(l1 * 1.7, params)))
    p1, l1, g1 = runs[0][0], runs[0][1], runs[0][2:]
    assert_close(p1, p0, rtol=2e-5, atol=1e-6, what="probs")
    assert_close(l1, l0, rtol=2e-6, atol=1e-6, what="loss")
    for a, r, n in zip(g1, g0, ("z", "W1", "b1", "W2", "b2")):
        assert_close(a, r, rtol=2e-4, atol=2e-7, what="d" + n)
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)
    # through the model: GraphMIL(labels=...) in training mode == forward + cross_entropy_from_probs


def test_graph_head_supported_bound_and_bad_label():
    """isic_graph_head_supported mirrors the kernel's own LDS / class bound (the entry returns ERR_UNSUPPORTED exactly where
    the query says 0), and a label outside [0, C) turns the fused loss into NaN instead of reading past the row's probabilities."""
    from isic_hip import ops
    from isic_hip.lib import IsicHipError
    assert ops.graph_head_supported(128, 128, 7) and ops.graph_head_supported(32, 24, 7)
    assert not ops.graph_head_supported(256, 128, 7) and not ops.graph_head_supported(128, 128, 16)
    gen = torch.Generator().manual_seed(2)
    mk = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    z, W1, b1, W2, b2 = mk(9, 256), mk(128, 256), mk(128), mk(7, 128), mk(7)
    y = torch.arange(9).to(DEV) % 7
    with pytest.raises(IsicHipError):
        ops.graph_head_loss(z, W1, b1, W2, b2, y)
    z, W1 = mk(9, 128), mk(128, 128)
    _, loss = ops.graph_head_loss(z, W1, b1, W2, b2, y)
    assert torch.isfinite(loss)
    y[4] = 7
    _, loss = ops.graph_head_loss(z, W1, b1, W2, b2, y)
    assert torch.isnan(loss)


@pytest.mark.parametrize("hidden", [128, 256])
def test_graphmil_train_step_with_labels_fused_head_and_fallback(hidden):
    """GraphMIL(labels=...) in training mode at the reference CLI's `--hidden-dim 256` (05_train_gnns.py:407; the fused head's
    LDS bound stops at ~136: the operator chain must take over) and at 128 (fused head): probabilities, loss and every parameter
    gradient equal the plain forward + cross_entropy_from_probs path with the same dropout words (fp32 summation-order tolerance)."""
    from gnn_models import GraphMIL
    from isic_hip import ops, train as T
    gen = torch.Generator().manual_seed(17)
    G, n, D, k = 12, 40, 48, 4
    recs = []
    for i in range(G):
        src = torch.arange(n).repeat_interleave(k)
        dst = (src + 1 + torch.randint(0, n - 1, (n * k,), generator=gen)) % n
        recs.append({"x": torch.randn(n, D, generator=gen), "edge_index": torch.stack([src, dst]), "y": i % 7})
    store = T.GraphStore(recs, torch.device(DEV), True, mode="gcn")
    torch.manual_seed(6)
    model = GraphMIL(input_dim=D, gnn_type="gcn", gnn_hidden=hidden, gnn_layers=2, gnn_dropout=0.5, att_dim=32, att_heads=4,
                     pool_dropout=0.2, classifier_dim=128, classifier_light=True, num_classes=7).to(DEV)
    model.train()
    assert ops.graph_head_supported(hidden, 128, 7) == (hidden == 128)
    idx = torch.arange(G).to(DEV)
    y = store.y_dev[idx]
    params = list(model.parameters())
    x, offs, g = store.batch(idx)

    def run(with_labels):
        model.set_dropout_state(seed=4, step=0)
        if with_labels:
            probs, _att, loss = model(x, offsets=offs, graph=g, labels=y)
        else:
            probs, _att = model(x, offsets=offs, graph=g)
            loss = ops.cross_entropy_from_probs(probs, y)
        return probs.detach().clone(), loss.detach().clone(), torch.autograd.grad(loss, params)
    pa, la, ga = run(False)
    pb, lb, gb = run(True)
    assert_close(pb, pa, rtol=2e-5, atol=1e-6, what="probs")
    assert_close(lb, la, rtol=2e-6, atol=1e-6, what="loss")
    for (name, _), u, v in zip(model.named_parameters(), gb, ga):
        assert_close(u, v, rtol=3e-4, atol=3e-7, what=name)


def test_graphmil_bench_geometry_whole_model_vs_oracle():
    """The configs[3] step AT THE GEOMETRY bench.py TIMES -- 668 k-NN graphs of 196 nodes x 768 features per launch, 3-layer
    GCN F = 128, 4 attention heads, light classifier, dropout 0.5 / 0.2 -- through the path the driver runs
    (GraphStore.batch_rows -> the input projection reading through the row index, the persistent / row-panel / A^T B fp32 GEMM
    kernels, GcnBlockFn, the fused head + loss, once eagerly and once as a replayed hipGraph with the device step clock) against
    oracle/gnn.py run graph by graph with the batch's dropout words: probabilities, per-graph loss, node embeddings of every
    graph and EVERY parameter gradient (the mean of the 256 per-graph oracle gradients).  fp32 summation-order tolerance."""
    from gnn_models import GraphMIL
    from isic_hip import graphs as G, ops, optim, train as T
    from isic_hip.bags import BagOffsets
    from isic_hip.graph import knn_indices
    dev = torch.device(DEV)
    Gs, N, D, F_, L, k, C = 668, 196, 768, 128, 3, 8, 7        # bench.py's --graphs-per-step default
    n_graphs = 700
    gen = torch.Generator().manual_seed(77)
    y = torch.arange(n_graphs) % C
    x = torch.randn(n_graphs, N, D, generator=gen) + 0.25 * y.view(-1, 1, 1).float()
    xd = x.to(dev)
    nn_idx = knn_indices(xd.view(-1, D), BagOffsets.from_lengths([N] * n_graphs, dev), k).view(n_graphs, N, k)
    src = torch.arange(N, device=dev).view(1, N, 1).expand(n_graphs, N, k)
    ei = torch.stack([src.reshape(n_graphs, -1), nn_idx.reshape(n_graphs, -1)], dim=1)
    records = [{"x": xd[i], "edge_index": ei[i], "y": int(y[i])} for i in range(n_graphs)]
    torch.manual_seed(42)
    model = GraphMIL(input_dim=D, gnn_type="gcn", gnn_hidden=F_, gnn_layers=L, gnn_dropout=0.5, gnn_heads=4, att_dim=128,
                     att_heads=4, pool_dropout=0.2, classifier_dim=128, classifier_light=True, num_classes=C).to(dev)
    model.train()
    opt = optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)       # flat gradient buffer: the kernels accumulate into it
    store = T.GraphStore(records, dev, True, mode=model.graph_mode)
    idx = torch.randint(0, n_graphs, (Gs,), generator=gen).to(dev)          # with repeats, as the bench draws them
    labels = store.y_dev[idx]

    def fwd_bwd():
        xb, rows, n_rows, ob, gb = store.batch_rows(idx)
        opt.zero_grad()
        with ops.fused_grad_accumulation():
            probs, _att, loss = model(xb, offsets=ob, graph=gb, labels=labels, x_rows=(rows, n_rows))
            ops.backward(loss)
        return probs, loss
    # ---- eager, host dropout clock at (seed 5, step 0)
    model.set_dropout_state(seed=5, step=0)
    probs, loss = fwd_bwd()
    torch.cuda.synchronize()
    got = {"probs": probs.detach().cpu(), "loss": float(loss.detach()), "emb": model.last_node_embeddings.cpu().clone(),
           "grad": {kk: p.grad.detach().cpu().clone() for kk, p in model.named_parameters()}}
    # ---- the same step as a replayed hipGraph (device clock at step 0; the warm-up is undone by CapturedStep)
    model.set_dropout_state(seed=5, step=0)
    clock = G.StepClock(dev).attach(model, None)
    cap = G.CapturedStep(lambda: fwd_bwd()[1], warmup=2)
    cap.replay()
    torch.cuda.synchronize()
    for kk, p in model.named_parameters():
        if kk.startswith("layer_norms"):           # dgamma / dbeta meet through fp32 atomics: last-bit noise
            assert_close(p.grad, got["grad"][kk], rtol=2e-5, atol=1e-8, what="captured " + kk)
        else:
            assert torch.equal(p.grad.cpu(), got["grad"][kk]), kk
    G.StepClock.detach(model, None)
    # ---- oracle, graph by graph, with the batch's dropout words
    cfg = dict(gnn_type="gcn", gnn_hidden=F_, gnn_layers=L, gnn_dropout=0.5, att_dim=128, classifier_dim=128, pool_dropout=0.2)
    p0 = {kk: v.detach().float().cpu() for kk, v in model.state_dict().items()}
    q = {kk: v.clone().requires_grad_(True) for kk, v in p0.items()}
    ei_c, idx_c = ei.cpu(), idx.cpu().tolist()
    tot, o_probs, o_emb, o_loss = 0.0, [], [], []
    for b, gi in enumerate(idx_c):
        out = gnn.graphmil_forward(q, cfg, x[gi], ei_c[gi], drop={"seed": 5, "stream_base": 0, "node_offset": b * N, "graph_index": b})
        l = gnn.graph_loss(out["probs"], int(y[gi]))
        (l / Gs).backward()
        o_probs.append(out["probs"].detach()); o_emb.append(out["hs"][-1].detach()); o_loss.append(float(l.detach()))
    assert_close(got["probs"], torch.stack(o_probs), rtol=1e-4, atol=2e-6, what="probs")
    per_graph = -torch.log(got["probs"][torch.arange(Gs), labels.cpu()] + 1e-9)
    assert_close(per_graph, torch.tensor(o_loss), rtol=2e-4, atol=2e-6, what="per-graph loss")
    assert abs(got["loss"] - float(np.mean(o_loss))) < 2e-5 * max(1.0, abs(float(np.mean(o_loss))))
    assert_close(got["emb"], torch.cat(o_emb), rtol=1e-4, atol=1e-5, what="node embeddings")
    for kk, prm in q.items():
        if kk.startswith("attention_layers") and kk.endswith("2.bias"):
            assert float(got["grad"][kk].abs().max()) < 1e-6          # analytically zero (softmax shift invariance)
            continue
        assert_close(got["grad"][kk], prm.grad, rtol=1e-3, atol=1e-7, what=kk)


def test_linear_rows_reads_through_the_index_like_gather_then_linear():
    """ops.linear_rows (isic_gemm_f32_rows_ws: the persistent GEMM reads its row operand through an int32 index, forward as A rows,
    weight gradient as the k rows of B) == gather + ops.linear bit for bit (same kernel, same summation order), on a batch of
    256 x 196 rows drawn with repeats from a 300-graph store; and GraphStore.batch_rows + GraphMIL(x_rows=...) == batch + GraphMIL."""
    from isic_hip import ops
    gen = torch.Generator().manual_seed(31)
    G, n, D, F = 300, 196, 768, 128
    store = torch.randn(G * n, D, generator=gen).to(DEV)
    sel = torch.randint(0, G, (256,), generator=gen)
    rows = (sel.view(-1, 1) * n + torch.arange(n).view(1, -1)).reshape(-1).to(torch.int32)
    M = rows.numel()
    rows_pad = torch.cat([rows, torch.zeros(8, dtype=torch.int32)]).to(DEV)
    W = (torch.randn(F, D, generator=gen) * 0.05).to(DEV).requires_grad_(True)
    b = torch.randn(F, generator=gen).to(DEV).requires_grad_(True)
    dy = torch.randn(M, F, generator=gen).to(DEV)
    y0 = ops.linear(store[rows.to(DEV).long()], W, b)
    g0 = torch.autograd.grad(y0, (W, b), dy)
    y1 = ops.linear_rows(store, rows_pad, M, W, b)
    g1 = torch.autograd.grad(y1, (W, b), dy)
    assert torch.equal(y1, y0), (y1 - y0).abs().max().item()
    assert torch.equal(g1[0], g0[0]), (g1[0] - g0[0]).abs().max().item()
    assert torch.equal(g1[1], g0[1])
    # a shape the gathering kernel does not take (small M) falls back to gather + linear
    y2 = ops.linear_rows(store, rows_pad, 64, W, b)
    assert_close(y2, y0[:64], rtol=2e-5, atol=2e-5, what="fallback")


def test_batch_rows_path_equals_gathered_batch():
    """GraphStore.batch_rows + GraphMIL(x_rows=...) -- the input projection and its weight gradient read the node features
    through the batch's row index (isic_csr_batch_assemble's row_index_out -> isic_gemm_f32_rows_ws) -- against GraphStore.batch +
    GraphMIL on the gathered features: same kernels, same order of summation -> bit-equal probabilities, loss and gradients."""
    from gnn_models import GraphMIL
    from isic_hip import train as T
    gen = torch.Generator().manual_seed(41)
    G, n, D, k = 80, 196, 768, 8
    recs = []
    for i in range(G):
        src = torch.arange(n).repeat_interleave(k)
        dst = (src + 1 + torch.randint(0, n - 1, (n * k,), generator=gen)) % n          # no self loops: equal entry counts per graph
        recs.append({"x": torch.randn(n, D, generator=gen), "edge_index": torch.stack([src, dst]), "y": i % 7})
    store = T.GraphStore(recs, torch.device(DEV), True, mode="gcn")
    torch.manual_seed(5)
    model = GraphMIL(input_dim=D, gnn_type="gcn", gnn_hidden=128, gnn_layers=2, gnn_dropout=0.3, att_dim=64, att_heads=2,
                     pool_dropout=0.2, classifier_dim=128, classifier_light=True, num_classes=7).to(DEV)
    model.train()
    idx = torch.randint(0, G, (64,), generator=gen).to(DEV)
    y = store.y_dev[idx]
    params = [p for p in model.parameters()]

    def run(rows_path):
        model.set_dropout_state(seed=9, step=0)
        if rows_path:
            xs, rows, nr, offs, g = store.batch_rows(idx)
            probs, att, loss = model(xs, offsets=offs, graph=g, labels=y, x_rows=(rows, nr))
        else:
            x, offs, g = store.batch(idx)
            probs, att, loss = model(x, offsets=offs, graph=g, labels=y)
        return probs.detach().clone(), att.detach().clone(), loss.detach().clone(), torch.autograd.grad(loss, params)
    pa, aa, la, ga = run(False)
    pb, ab, lb, gb = run(True)
    assert torch.equal(pa, pb) and torch.equal(aa, ab) and torch.equal(la, lb)
    for (name, _), u, v in zip(model.named_parameters(), ga, gb):
        assert torch.equal(u, v), (name, (u - v).abs().max().item())
