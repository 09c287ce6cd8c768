"""Generate ``tests/golden/*.npz`` by running the REFERENCE's own pure-torch code
(oracle / test infrastructure; runs in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [/root/reference]

The reference is imported from where it lies (nothing is copied into this
repo); packages it imports at module level but that are absent offline
(`ray`, `torch_geometric`, `neptune`, `umap`, `efficientnet_pytorch`, `cv2`,
`torchvision`, `albumentations`) are replaced by empty stand-in modules --
none of their code is needed for the functions exercised here.  Inputs and
weights are closed-form (``oracle/formula.py``), so fixtures hold only small
outputs (full tensors for small cases, strided samples + norms for big grads).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
from collections import OrderedDict

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import formula  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    ray = _stub("ray")
    ray.tune = _stub("ray.tune", report=lambda *a, **k: None)
    pyg = _stub("torch_geometric")
    pyg.nn = _stub("torch_geometric.nn")
    _stub("neptune")
    _stub("umap")

    class _EffStub:
        @staticmethod
        def from_pretrained(name):  # nothing is fetched: a local empty module
            return nn.Identity()

    _stub("efficientnet_pytorch", EfficientNet=_EffStub)


def load_ref(fname, modname, tolerate=()):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    try:
        spec.loader.exec_module(mod)
    except tolerate:
        pass  # module-level driver code on hard-coded paths; functions are defined by then
    return mod


def np_(t):
    return t.detach().cpu().numpy().copy()  # copy: state_dict tensors are updated in place later


def sample_stats(t, maxn=2048):
    f = np_(t).reshape(-1).astype(np.float64)
    stride = max(1, f.size // maxn)
    return {"sample": f[::stride][:maxn].astype(np.float32), "stride": np.int64(stride),
            "sum": np.float64(f.sum()), "l2": np.float64(np.sqrt((f * f).sum())), "numel": np.int64(f.size)}


def names_blob(shapes):
    return np.array([f"{k}|{','.join(map(str, v))}" for k, v in shapes.items()])


def put_grads(d, prefix, named_grads, full):
    for k, g in named_grads.items():
        if full:
            d[f"{prefix}{k}"] = np_(g)
        else:
            for kk, vv in sample_stats(g).items():
                d[f"{prefix}{k}#{kk}"] = vv


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)  # 05 opens config.yml from CWD (05_train_gnns.py:36-39)
    sys.argv = ["gen_golden"]
    try:
        ugm = load_ref("utils_g_mil.py", "ref_utils_g_mil")
        g03 = load_ref("03_build_graphs.py", "ref_build_graphs", tolerate=(FileNotFoundError, OSError))
        g05 = load_ref("05_train_gnns.py", "ref_train_gnns")
        mdl = load_ref("model.py", "ref_model")
    finally:
        os.chdir(cwd)
    torch.manual_seed(0)
    torch.set_num_threads(1)

    # ---------------- AttentionMIL_teacher / AttentionMIL (utils_g_mil.py:15-105)
    cases = [("small", 16, 32, 16, 8, 3, True), ("ref", 196, 768, 128, 64, 7, False),
             ("tuned", 196, 768, 368, 772, 7, False),
             ("rag1", 1, 32, 16, 8, 3, True), ("rag5", 5, 32, 16, 8, 3, True),
             ("rag64", 64, 32, 16, 8, 3, True), ("rag196", 196, 32, 16, 8, 3, True)]
    for tag, N, D, H, A, C, full in cases:
        m = ugm.AttentionMIL_teacher(input_dim=D, hidden_dim=H, att_dim=A, dropout=0.5, num_classes=C)
        shapes = formula.shapes_of(m)
        m.load_state_dict(formula.formula_state_dict(shapes))
        m.eval()
        x = formula.formula_input(N, D).requires_grad_(True)
        y = torch.tensor([N % C])
        out = m(x)
        loss = nn.CrossEntropyLoss()(out["bag_logits"].unsqueeze(0), y)  # 01:143,244
        loss.backward()
        d = {"names": names_blob(shapes), "dims": np.array([N, D, H, A, C]), "label": np_(y), "loss": np_(loss)}
        for k, v in out.items():
            d[f"out.{k}"] = np_(v)
        grads = OrderedDict((k, p.grad) for k, p in m.named_parameters())
        grads["x"] = x.grad
        put_grads(d, "grad.", grads, full)
        np.savez_compressed(os.path.join(OUT, f"teacher_{tag}.npz"), **d)

        m2 = ugm.AttentionMIL(input_dim=D, hidden_dim=H, att_dim=A, dropout=0.5, num_classes=C)
        shapes2 = formula.shapes_of(m2)
        m2.load_state_dict(formula.formula_state_dict(shapes2))
        m2.eval()
        probs, a = m2(formula.formula_input(N, D))
        np.savez_compressed(os.path.join(OUT, f"attmil_{tag}.npz"), names=names_blob(shapes2),
                            dims=np.array([N, D, H, A, C]), probs=np_(probs), a=np_(a))

    # ---------------- 3 AdamW steps of the 01 loop on 8 bags (01:217-246)
    N, D, H, A, C = 12, 32, 16, 8, 3
    m = ugm.AttentionMIL_teacher(input_dim=D, hidden_dim=H, att_dim=A, dropout=0.0, num_classes=C)
    shapes = formula.shapes_of(m)
    m.load_state_dict(formula.formula_state_dict(shapes))
    m.train()
    bags = [formula.formula_input(N, D, phase=0.5 + 0.3 * i) for i in range(8)]
    labels = [i % C for i in range(8)]
    ds = ugm.PatientDataset([b.numpy() for b in bags], labels)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    opt = torch.optim.AdamW(m.parameters(), lr=2.2e-4 * 50, weight_decay=8.6e-4)
    crit = nn.CrossEntropyLoss()
    d = {"names": names_blob(shapes), "dims": np.array([N, D, H, A, C]), "lr": 2.2e-4 * 50, "wd": 8.6e-4}
    step = 0
    for x, yb in loader:
        x = x[0]
        yl = yb.long()
        opt.zero_grad()
        out = m(x)
        loss = crit(out["bag_logits"].unsqueeze(0), yl)
        loss.backward()
        opt.step()
        d[f"loss{step}"] = np_(loss)
        for k, v in m.state_dict().items():
            d[f"step{step}.{k}"] = np_(v)
        step += 1
        if step == 3:
            break
    np.savez_compressed(os.path.join(OUT, "teacher_adamw3.npz"), **d)

    # ---------------- graph builders (03_build_graphs.py:15-78, utils_g_mil.py:564-674)
    d = {"grid4": np_(g03._grid_edge_index(False)), "grid8": np_(g03._grid_edge_index(True))}
    for diag in (False, True):
        adj_norm, adj_mask, ei, ew = ugm.build_graph(torch.zeros(196, 4), "grid", connect_diagonals=diag)
        d[f"gridadj{int(diag)}.edge_index"] = np_(ei)
        d[f"gridadj{int(diag)}.edge_weight"] = np_(ew)
    for seed, r in ((42, 4), (10042, 1), (20049, 16)):
        d[f"random.{seed}.{r}"] = np_(g03._random_edge_index(196, r=r, seed=seed))
    d["random.small"] = np_(g03._random_edge_index(7, r=3, seed=5))
    for tag, n, dd in (("a", 196, 768), ("b", 64, 512), ("c", 17, 8)):
        xg = formula.gapped_points(n, dd, seed=n)
        for k in (1, 3, 8, 16):
            d[f"knn.{tag}.{k}"] = np_(g03._knn_edge_index(xg, k))
        d[f"knnu.{tag}.8"] = np_(ugm.build_knn_edge_index(xg, 8))
    d["knn.tiny1"] = np_(g03._knn_edge_index(torch.zeros(1, 4), 3))
    d["knn.clampk"] = np_(g03._knn_edge_index(formula.gapped_points(5, 4, seed=5), 99))
    np.savez_compressed(os.path.join(OUT, "graphs.npz"), **d)

    # ---------------- GraphMIL[mlp] (05_train_gnns.py:51-219), 05 call-site config
    for tag, N, D, F_, L in (("small", 16, 24, 16, 2), ("ref", 196, 768, 128, 2), ("same", 20, 16, 16, 3)):
        gm = g05.GraphMIL(input_dim=D, gnn_type="mlp", gnn_hidden=F_, gnn_layers=L, gnn_dropout=0.5,
                          gnn_heads=4, gnn_concat=True, att_dim=(128 if tag == "ref" else 8), att_heads=4,
                          pool_dropout=0.2, classifier_dim=(128 if tag == "ref" else 12),
                          classifier_light=True, num_classes=7, use_residual=True, use_layer_norm=True)
        shapes = formula.shapes_of(gm)
        gm.load_state_dict(formula.formula_state_dict(shapes))
        gm.eval()
        hs = []
        hooks = []
        # node embeddings h after each layer = input of the next layer / of the pool
        for i in range(1, L):
            hooks.append(gm.gnn_layers[i].register_forward_pre_hook(lambda mod, inp: hs.append(inp[0].detach().clone())))
        hooks.append(gm.attention_layers[0].register_forward_pre_hook(lambda mod, inp: hs.append(inp[0].detach().clone())))
        x = formula.formula_input(N, D).requires_grad_(True)
        probs, att = gm(x, None)
        y = torch.tensor([3])
        loss = nn.CrossEntropyLoss()(torch.log(probs + 1e-9).unsqueeze(0), y)  # 05:344
        loss.backward()
        d = {"names": names_blob(shapes), "dims": np.array([N, D, F_, L]), "probs": np_(probs), "att": np_(att),
             "loss": np_(loss), "label": np_(y),
             "att_dim": np.int64(128 if tag == "ref" else 8), "classifier_dim": np.int64(128 if tag == "ref" else 12)}
        for i, h in enumerate(hs):
            d[f"h{i}"] = np_(h)
        grads = OrderedDict((k, p.grad) for k, p in gm.named_parameters())
        grads["x"] = x.grad
        put_grads(d, "grad.", grads, tag != "ref")
        for hk in hooks:
            hk.remove()
        np.savez_compressed(os.path.join(OUT, f"graphmil_mlp_{tag}.npz"), **d)

    # ---------------- heterophily measures (04_measure_heterophily.py:107-169), the reference's own numpy code
    h04 = load_ref("04_measure_heterophily.py", "ref_heterophily", tolerate=(FileNotFoundError, OSError))
    Nn, Dd, Cc = 196, 48, 7
    emb = formula.formula_input(Nn, Dd, phase=0.4).numpy()
    logits = formula.formula_input(Nn, Cc, phase=1.1) * 3.0
    pp = torch.softmax(logits, dim=1).numpy()
    row = {"patch_embeddings": emb, "patch_probs": pp, "dominant_class": pp.argmax(axis=1).astype(np.int32),
           "grid4_edge_index": g03._grid_edge_index(False).numpy(), "grid8_edge_index": g03._grid_edge_index(True).numpy(),
           "knn_edge_indices": {k: g03._knn_edge_index(torch.from_numpy(emb), k).numpy() for k in (3, 8)},
           "random_edge_indices": {r: g03._random_edge_index(Nn, r, 42 + r).numpy() for r in (2,)}}

    class _Row:                      # the reference reads attributes of an itertuples() row
        def __init__(self, d):
            self.__dict__.update(d)
    # one variant with self loops and a duplicated edge mixed in: the reference strips loops before averaging
    loops = np.stack([np.arange(0, Nn, 7), np.arange(0, Nn, 7)])
    row["edge_index"] = np.concatenate([row["knn_edge_indices"][3][:, :300], loops, row["knn_edge_indices"][3][:, :5]], axis=1)
    d = {"N": np.int64(Nn), "D": np.int64(Dd), "C": np.int64(Cc), "edge_index": row["edge_index"]}
    for variant in (None, "grid4", "grid8", "knn3", "knn8", "random2"):
        em = h04.compute_edge_heterophily(_Row(row), graph_variant=variant)
        tag = variant or "raw"
        for k in ("H_kl", "H_dirichlet", "H_spatial", "lambda_2", "H_compat_matrix"):
            d[f"{tag}.{k}"] = np.asarray(em[k], dtype=np.float64)
        d[f"{tag}.H_adj"] = np.float64(em["H_adj"])
        sm = h04._summarize_image(em, {})
        for k in ("num_edges", "H_kl_mean", "H_kl_std", "H_kl_median", "H_dirichlet_mean", "H_spatial_median", "H_adj_mean",
                  "lambda_2_mean"):
            d[f"{tag}.sum.{k}"] = np.float64(sm[k])
    np.savez_compressed(os.path.join(OUT, "heterophily.npz"), **d)

    # ---------------- net_utils.train / validate / test / EarlyStopping (net_utils.py:6-158)
    # The reference's own loops drive the reference's own MultiModalFusionNet (non-image modalities) on CPU:
    # two SGD epochs over three dict batches (dropout probabilities set to 0 so that no RNG stream is involved),
    # then validate / test on a fourth batch, and EarlyStopping over a fixed loss sequence.
    import contextlib
    import io
    nu = load_ref("net_utils.py", "ref_net_utils")
    R, B = 32, 8
    net = mdl.MultiModalFusionNet(modality=["radiomics", "clinical", "artifacts"], fusion_level="intermediate",
                                  fusion_strategy="concat", radiomics_dim=R)
    shapes = formula.shapes_of(net)
    net.load_state_dict(formula.formula_state_dict(shapes))
    for mod in net.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0

    def make_batch(i):
        return {"image": torch.zeros(B, 1), "radiomics": formula.formula_input(B, R, phase=0.3 + 0.5 * i),
                "age": formula.ftensor((B,), 0.5, 0.3, 0.1 + i), "sex": (torch.arange(B) + i) % 3,
                "loc": (torch.arange(B) * 2 + i) % 15, "artifacts": ((torch.arange(B * 6).view(B, 6) + i) % 2),
                "target": (torch.arange(B) * 3 + i) % 7}

    loader = [make_batch(i) for i in range(3)]
    held = [make_batch(5)]
    crit = nn.CrossEntropyLoss()
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    d = {"names": names_blob(shapes), "B": np.int64(B), "R": np.int64(R)}
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        for ep in range(2):
            nu.train(net, loader, crit, opt, "cpu", None, ep)
        d["val_loss"] = np.float64(nu.validate(net, held, crit, "cpu", None, 2))
        acc, report = nu.test(net, held, "cpu", None)
    d["test_acc"] = np.float64(acc)
    d["report"] = np.array(report)
    d["stdout"] = np.array(buf.getvalue())
    for k, v in net.state_dict().items():
        if k.startswith(("image_model", "image_proj")):
            continue                       # the image branch is not part of this model instance's forward
        if v.numel() <= 4096:
            d[f"after.{k}"] = np_(v)
        else:
            for kk, vv in sample_stats(v, 512).items():
                d[f"after.{k}#{kk}"] = vv
    es = nu.EarlyStopping(patience=3, neptune_run=None)
    seq = [1.0, 0.8, 0.9, 0.85, 0.7, 0.71, 0.72, 0.73]
    flags, counters, bests = [], [], []
    for l in seq:
        flags.append(bool(es(l, net)))
        counters.append(es.counter)
        bests.append(es.best_loss)
    d["es_losses"], d["es_flags"], d["es_counters"], d["es_best"] = (np.array(seq), np.array(flags), np.array(counters),
                                                                      np.array(bests))
    np.savez_compressed(os.path.join(OUT, "net_utils_loops.npz"), **d)

    # ---------------- radiomics_mlp / AttentionFusion(_Late) / fusion branches (model.py:6-40, 206-227)
    # forward logits AND the gradient of CE(logits, target) w.r.t. every parameter the forward touches, for both
    # fusion levels ("late": per-modality heads model.py:155-164, sum / softmax-weighted / AttentionFusion_Late :216-227)
    for level in ("intermediate", "late"):
        for R in (32, 128):
            for strat in ("concat", "weighted", "attention"):
                net = mdl.MultiModalFusionNet(modality=["radiomics", "clinical", "artifacts"],
                                              fusion_level=level, fusion_strategy=strat, radiomics_dim=R)
                shapes = formula.shapes_of(net)
                net.load_state_dict(formula.formula_state_dict(shapes))
                net.eval()
                B = 6
                rad = formula.formula_input(B, R, phase=0.9)
                age = formula.ftensor((B,), 0.5, 0.3, 0.1)
                sex = torch.arange(B) % 3
                loc = torch.arange(B) % 15
                art = (torch.arange(B * 6).view(B, 6) % 2)
                target = (torch.arange(B) * 3 + 1) % 7
                net.zero_grad()
                logits = net(None, rad, age, sex, loc, art)
                loss = nn.CrossEntropyLoss()(logits, target)
                loss.backward()
                rf = net.radiomics_mlp(rad)
                d = {"names": names_blob(shapes), "logits": np_(logits), "rad_feat": np_(rf), "B": np.int64(B),
                     "R": np.int64(R), "target": np_(target), "loss": np_(loss)}
                for k, prm in net.named_parameters():
                    if prm.grad is None:
                        continue                      # image branch: not part of this forward
                    put_grads(d, "grad.", {k: prm.grad}, full=prm.numel() <= 4096)
                if strat == "attention" and level == "intermediate":
                    feats = [rf, formula.formula_input(B, 128, phase=1.7), formula.formula_input(B, 128, phase=2.9)]
                    d["attfusion"] = np_(net.attention(feats))
                if strat == "attention" and level == "late":
                    lg = [formula.formula_input(B, 7, phase=1.7 + 1.2 * i) for i in range(3)]
                    d["attfusion_late"] = np_(net.attention(lg))
                name = f"fusion_{strat}_R{R}.npz" if level == "intermediate" else f"fusion_late_{strat}_R{R}.npz"
                np.savez_compressed(os.path.join(OUT, name), **d)

    # ---------------- metrics (sklearn; 05:290,299)
    from sklearn.metrics import balanced_accuracy_score, roc_auc_score
    rng = np.random.RandomState(7)
    y = np.arange(64) % 7
    rng.shuffle(y)
    s = rng.rand(64, 7) + 0.8 * np.eye(7)[y]
    s[5] = s[9]  # ties
    s = s / s.sum(axis=1, keepdims=True)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), y=y, scores=s,
                        auc=roc_auc_score(y, s, multi_class="ovr", labels=np.arange(7)),
                        bacc=balanced_accuracy_score(y, s.argmax(axis=1)))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
