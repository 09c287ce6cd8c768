"""Pins the CPU oracle against golden vectors produced by the REFERENCE's own
code (``oracle/gen_golden.py``).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import formula, fusion, gnn, graphs, metrics, mil
from helpers import assert_close, check_grad, formula_params, load_golden

TEACHER_CASES = ["small", "ref", "tuned", "rag1", "rag5", "rag64", "rag196"]


@pytest.mark.parametrize("tag", TEACHER_CASES)
def test_teacher_forward_backward(tag):
    g = load_golden(f"teacher_{tag}.npz")
    N, D, H, A, C = (int(v) for v in g["dims"])
    p = formula_params(g)
    x = formula.formula_input(N, D)
    loss, out, grads = mil.teacher_loss_and_grads(p, x, torch.from_numpy(g["label"]))
    for k in ("bag_logits", "bag_probs", "attention", "patch_logits", "patch_probs"):
        assert_close(out[k], g[f"out.{k}"], rtol=2e-5, atol=1e-6, what=k)
    assert_close(loss, g["loss"], rtol=2e-5, what="loss")
    for k in list(p) + ["x"]:
        check_grad(g, k, grads[k])


@pytest.mark.parametrize("tag", TEACHER_CASES)
def test_attention_mil(tag):
    g = load_golden(f"attmil_{tag}.npz")
    N, D, H, A, C = (int(v) for v in g["dims"])
    probs, a, _, _ = mil.attention_mil_forward(formula_params(g), formula.formula_input(N, D))
    assert_close(probs, g["probs"], rtol=2e-5, what="probs")
    assert_close(a, g["a"], rtol=2e-5, what="a")


def test_teacher_adamw_three_steps():
    g = load_golden("teacher_adamw3.npz")
    N, D, H, A, C = (int(v) for v in g["dims"])
    p = formula_params(g)
    bags = [formula.formula_input(N, D, phase=0.5 + 0.3 * i) for i in range(8)]
    labels = [i % C for i in range(8)]
    hist = mil.per_bag_train_loop(p, bags, labels, 3, float(g["lr"]), float(g["wd"]))
    for s, (params, loss) in enumerate(hist):
        assert abs(loss - float(g[f"loss{s}"])) < 1e-5
        for k, v in params.items():
            assert_close(v, g[f"step{s}.{k}"], rtol=1e-5, atol=1e-7, what=f"step{s}.{k}")


def test_batched_equals_per_bag():
    g = load_golden("teacher_small.npz")
    p = formula_params(g)
    lens = [3, 1, 7, 5]
    offs = np.concatenate([[0], np.cumsum(lens)])
    x = formula.formula_input(int(offs[-1]), 32)
    out = mil.teacher_forward_batched(p, x, offs)
    for b in range(len(lens)):
        o = mil.teacher_forward(p, x[offs[b]:offs[b + 1]])
        assert_close(out["bag_logits"][b], o["bag_logits"], rtol=1e-6)
        assert_close(out["attention"][offs[b]:offs[b + 1]], o["attention"], rtol=1e-6)


def test_graph_builders():
    g = load_golden("graphs.npz")
    assert np.array_equal(graphs.grid_edge_index(14, False).numpy(), g["grid4"])
    assert np.array_equal(graphs.grid_edge_index(14, True).numpy(), g["grid8"])
    for diag in (False, True):
        _, _, ei, ew = graphs.build_graph(torch.zeros(196, 4), "grid", connect_diagonals=diag)
        assert np.array_equal(ei.numpy(), g[f"gridadj{int(diag)}.edge_index"])
        assert_close(ew, g[f"gridadj{int(diag)}.edge_weight"], rtol=1e-6)
    for seed, r in ((42, 4), (10042, 1), (20049, 16)):
        assert np.array_equal(graphs.random_edge_index(196, r, seed).numpy(), g[f"random.{seed}.{r}"])
    assert np.array_equal(graphs.random_edge_index(7, 3, 5).numpy(), g["random.small"])
    for tag, n, dd in (("a", 196, 768), ("b", 64, 512), ("c", 17, 8)):
        xg = formula.gapped_points(n, dd, seed=n)
        for k in (1, 3, 8, 16):
            assert np.array_equal(graphs.knn_edge_index(xg, k).numpy(), g[f"knn.{tag}.{k}"]), (tag, k)
        assert np.array_equal(graphs.build_graph(xg, "knn", k=8)[2].numpy(), g[f"knnu.{tag}.8"])
    assert graphs.knn_edge_index(torch.zeros(1, 4), 3).shape == tuple(g["knn.tiny1"].shape) == (2, 0)
    assert np.array_equal(graphs.knn_edge_index(formula.gapped_points(5, 4, seed=5), 99).numpy(), g["knn.clampk"])


@pytest.mark.parametrize("tag", ["small", "ref", "same"])
def test_graphmil_mlp(tag):
    g = load_golden(f"graphmil_mlp_{tag}.npz")
    N, D, F_, L = (int(v) for v in g["dims"])
    cfg = dict(gnn_type="mlp", gnn_hidden=F_, gnn_layers=L, att_dim=int(g["att_dim"]),
               classifier_dim=int(g["classifier_dim"]))
    p = formula_params(g)
    assert list(gnn.graphmil_shapes(D, cfg).items()) == [(k, tuple(v.shape)) for k, v in p.items()]
    loss, out, grads = gnn.graphmil_loss_and_grads(p, cfg, formula.formula_input(N, D), None, int(g["label"][0]))
    assert_close(out["probs"], g["probs"], rtol=2e-5, what="probs")
    assert_close(out["att"], g["att"], rtol=2e-5, what="att")
    assert_close(loss, g["loss"], rtol=2e-5, what="loss")
    for i in range(L):
        assert_close(out["hs"][i], g[f"h{i}"], rtol=2e-5, atol=2e-6, what=f"h{i}")
    for k in list(p) + ["x"]:
        check_grad(g, k, grads[k])


@pytest.mark.parametrize("level", ["intermediate", "late"])
@pytest.mark.parametrize("R", [32, 128])
@pytest.mark.parametrize("strat", ["concat", "weighted", "attention"])
def test_fusion(R, strat, level):
    """oracle/fusion.py vs the reference's own MultiModalFusionNet (model.py:166-227), both fusion levels: logits, the
    CE loss and the gradient of every parameter the forward touches."""
    g = load_golden(f"fusion_{strat}_R{R}.npz" if level == "intermediate" else f"fusion_late_{strat}_R{R}.npz")
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in formula_params(g).items()}
    B = int(g["B"])
    rad = formula.formula_input(B, R, phase=0.9)
    rf = fusion.mlp_ln_relu(p, "radiomics_mlp", rad)
    assert_close(rf, g["rad_feat"], rtol=2e-5, atol=2e-6, what="rad_feat")
    # clinical / artifact branches, model.py:186-204
    age = formula.ftensor((B,), 0.5, 0.3, 0.1)
    sex, loc = torch.arange(B) % 3, torch.arange(B) % 15
    art = torch.arange(B * 6).view(B, 6) % 2
    clin = torch.cat([age.unsqueeze(1), p["sex_emb.weight"][sex], p["loc_emb.weight"][loc]], dim=1)
    cf = fusion.mlp_ln_relu(p, "clinical_mlp", clin)
    af = fusion.mlp_ln_relu(p, "artifact_mlp",
                            torch.cat([p[f"artifact_embeddings.{i}.weight"][art[:, i]] for i in range(6)], dim=1))
    if level == "intermediate":
        logits = fusion.intermediate_fusion(p, [rf, cf, af], strat)
    else:
        logits = fusion.late_fusion(p, [rf, cf, af], ["radiomics", "clinical", "artifacts"], strat)
    assert_close(logits, g["logits"], rtol=5e-5, atol=5e-6, what="logits")
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(g["target"]))
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    loss.backward()
    keys = sorted({k[5:].split("#")[0] for k in g.files if k.startswith("grad.")})
    assert keys and all(k in p for k in keys)
    for k in keys:
        check_grad(g, k, p[k].grad if p[k].grad is not None else torch.zeros_like(p[k]))
    for k, v in p.items():          # what the reference's backward did not reach, the oracle's must not either
        if k not in keys and v.dtype.is_floating_point:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
    if strat == "attention" and level == "intermediate":
        feats = [rf, formula.formula_input(B, 128, phase=1.7), formula.formula_input(B, 128, phase=2.9)]
        assert_close(fusion.attention_fusion(p, feats)[0], g["attfusion"], rtol=2e-5, atol=2e-6)
    if strat == "attention" and level == "late":
        lg = [formula.formula_input(B, 7, phase=1.7 + 1.2 * i) for i in range(3)]
        assert_close(fusion.attention_fusion_late(p, lg), g["attfusion_late"], rtol=2e-5, atol=2e-6)


def test_metrics():
    g = load_golden("metrics.npz")
    assert abs(metrics.roc_auc_ovr_macro(g["y"], g["scores"]) - float(g["auc"])) < 1e-12
    assert abs(metrics.balanced_accuracy(g["y"], g["scores"].argmax(axis=1)) - float(g["bacc"])) < 1e-12
    with pytest.raises(ValueError):
        metrics.roc_auc_ovr_macro(np.zeros(8, dtype=int), np.full((8, 7), 1 / 7))


def test_heterophily_oracle_matches_reference_golden():
    """oracle/heterophily.py vs the reference's own `compute_edge_heterophily` (04_measure_heterophily.py:107-169) on
    a 196-node image: raw edge list with self loops and duplicated edges, both grids, k-NN and random graphs."""
    from oracle import formula, heterophily
    from oracle import graphs as ograph
    g = load_golden("heterophily.npz")
    N, D, C = int(g["N"]), int(g["D"]), int(g["C"])
    emb = formula.formula_input(N, D, phase=0.4).numpy()
    pp = torch.softmax(formula.formula_input(N, C, phase=1.1) * 3.0, dim=1).numpy()
    dom = pp.argmax(axis=1).astype(np.int32)
    variants = {"raw": g["edge_index"], "grid4": ograph.grid_edge_index(14, False), "grid8": ograph.grid_edge_index(14, True),
                "knn3": ograph.knn_edge_index(torch.from_numpy(emb), 3), "knn8": ograph.knn_edge_index(torch.from_numpy(emb), 8),
                "random2": ograph.random_edge_index(N, 2, 44)}
    for tag, ei in variants.items():
        em = heterophily.edge_heterophily(emb, pp, dom, np.asarray(ei))
        for k in ("H_kl", "H_dirichlet", "H_spatial", "H_compat_matrix"):
            np.testing.assert_allclose(em[k], g[f"{tag}.{k}"], rtol=1e-5, atol=1e-6, err_msg=f"{tag}.{k}")
        assert abs(em["H_adj"] - float(g[f"{tag}.H_adj"])) < 1e-9, tag
        # the reference's adjacency is float32 (04:150): a zero eigenvalue comes out as ~1e-8 noise there
        np.testing.assert_allclose(em["lambda_2"], g[f"{tag}.lambda_2"], rtol=1e-5, atol=1e-6, err_msg=f"{tag}.lambda_2")
