"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol that
include/isic_hip.h declares (no compute calls without a GPU), the header parser, bag offsets,
counter-based dropout bookkeeping, graph builders that live on the host, and that the product
refuses CPU tensors instead of silently falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import load_golden
from oracle import philox

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from isic_hip import lib
    L = lib.lib()
    text = open(os.path.join(ROOT, "include", "isic_hip.h")).read()
    public = set(re.findall(r"\b(isic_\w+)\s*\(", text))
    assert public == L.public, public ^ L.public
    assert not any("debug" in n or "test" in n for n in public)      # scaffolding stays out of the drop-in ABI
    ttext = open(os.path.join(ROOT, "include", "isic_hip_test.h")).read()
    declared = public | set(re.findall(r"\b(isic_test_\w+)\s*\(", ttext))
    assert declared == set(L.protos), declared ^ set(L.protos)
    assert len(public) >= 35
    cdll = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(cdll, name), name
    assert L.fn["isic_abi_version"]() == 1
    assert L.fn["isic_target_arch"]() == b"gfx950"
    # argument validation happens before any device work: usable without a GPU
    assert L.fn["isic_gcn_csr_workspace_bytes"](10, 20) == (6 * 10 + 2 * 30 + 20 + 64) * 4
    assert L.fn["isic_gemm_f32"](0, 0, -1, 4, 4, None, 4, None, 4, None, 4, None, 0, 0.0, None) == -1
    assert L.fn["isic_conv2d_igemm_bf16"](1, 1, 1, 1, 8, 8, 48, 8, 8, 64, 3, 3, 1, 1, 1, None, None, None, 0, None) == -2   # Cin % 64
    assert L.fn["isic_conv2d_wgrad_workspace_bytes"](2, 64, 7, 7, 64, 3, 3) >= 2 * 49 * 8


def test_header_prototypes_parse():
    from isic_hip import lib
    protos = lib.parse_header()
    rt, args = protos["isic_adam_step"]
    assert rt is ctypes.c_int and [a[1] for a in args][-1] == "stream"
    assert [a[0] for a in args][:5] == [ctypes.c_void_p] * 4 + [ctypes.c_int64]
    assert protos["isic_gcn_csr_workspace_bytes"][0] is ctypes.c_size_t
    with pytest.raises(TypeError):
        lib.call("isic_colsum_f32", 1, 2)


def test_cpu_tensors_are_rejected_everywhere():
    import utils_g_mil
    from gnn_models import GraphMIL
    from isic_hip.lib import IsicHipError
    from model import MultiModalMILNet
    with pytest.raises(IsicHipError):
        utils_g_mil.AttentionMIL(8, 4, 4, 0.0, 3)(torch.zeros(5, 8))
    with pytest.raises(IsicHipError):
        GraphMIL(8, "mlp", 8, 1, 0.0, att_dim=4, classifier_dim=4, classifier_light=True)(torch.zeros(5, 8))
    with pytest.raises(IsicHipError):
        MultiModalMILNet(hidden_dim=8, att_dim=4, radiomics_dim=4, encoder_layers=((64, 1),))(torch.zeros(2, 2, 3, 32, 32), torch.zeros(2, 4))
    with pytest.raises(ValueError):                      # 05:113-114
        GraphMIL(8, "sgc")
    GraphMIL(64, "fagcn", 64, 2)                         # every type the reference lists is built
    assert [k for k in GraphMIL(8, "gin", 8, 1).state_dict() if k.startswith("gnn_layers")] == [
        "gnn_layers.0.eps", "gnn_layers.0.nn.0.weight", "gnn_layers.0.nn.0.bias", "gnn_layers.0.nn.2.weight",
        "gnn_layers.0.nn.2.bias"]
    with pytest.raises(ValueError):
        GraphMIL(8, "nope")


def test_state_dict_surface_matches_reference_names():
    import utils_g_mil
    from gnn_models import GraphMIL
    from helpers import shapes_from_blob
    for fixture, make in (
        ("teacher_ref.npz", lambda: utils_g_mil.AttentionMIL_teacher(768, 128, 64, 0.5, 7)),
        ("attmil_ref.npz", lambda: utils_g_mil.AttentionMIL(768, 128, 64, 0.5, 7)),
        ("graphmil_mlp_ref.npz", lambda: GraphMIL(768, "mlp", 128, 2, 0.5, att_dim=128, att_heads=4, pool_dropout=0.2,
                                                   classifier_dim=128, classifier_light=True)),
    ):
        want = shapes_from_blob(load_golden(fixture)["names"])
        got = {k: tuple(v.shape) for k, v in make().state_dict().items()}
        assert list(got.items()) == list(want.items()), fixture


def test_dropout_spec_matches_oracle_definition():
    from isic_hip.ops import DropoutSpec
    for p in (0.0, 0.2, 0.5, 0.71, 0.999):
        s = DropoutSpec(p, seed=5, stream=9)
        assert s.threshold == (philox.dropout_threshold(p) if p > 0 else 0)
        if p > 0:
            assert s.scale == float(philox.dropout_scale(p))
    with pytest.raises(ValueError):
        DropoutSpec(1.0)
    # known-answer: Philox4x32-10 of the all-zero counter/key (Random123 reference vector)
    z = np.zeros(1, dtype=np.uint32)
    r = philox.philox4x32_10(z, z, z, z, 0, 0)
    assert [int(v[0]) for v in r] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]


def test_bag_offsets_and_sharding():
    from isic_hip.bags import BagOffsets
    from isic_hip.ddp import shard_range
    o = BagOffsets.from_lengths([3, 0, 5], "cpu")
    assert o.num_bags == 3 and o.max_bag == 5 and o.total == 8 and o.host.tolist() == [0, 3, 3, 8]
    assert BagOffsets.uniform(4, 7, "cpu").host.tolist() == [0, 7, 14, 21, 28]
    with pytest.raises(ValueError):
        BagOffsets([1, 2], "cpu")
    for n, w in ((10, 4), (3, 8), (64, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_host_side_graph_builders_golden():
    import build_graphs as bg
    import utils_g_mil
    g = load_golden("graphs.npz")
    assert np.array_equal(bg._grid_edge_index(False).numpy(), g["grid4"])
    assert np.array_equal(bg._grid_edge_index(True).numpy(), g["grid8"])
    for seed, r in ((42, 4), (10042, 1), (20049, 16)):
        assert np.array_equal(bg._random_edge_index(196, r, seed).numpy(), g[f"random.{seed}.{r}"])
    for diag in (False, True):
        _, _, ei, ew = utils_g_mil.build_graph(torch.zeros(196, 4), "grid", connect_diagonals=diag)
        assert np.array_equal(ei.numpy(), g[f"gridadj{int(diag)}.edge_index"])
        np.testing.assert_allclose(ew.numpy(), g[f"gridadj{int(diag)}.edge_weight"], rtol=1e-6)
    with pytest.raises(ValueError):
        utils_g_mil.build_grid_adj(10)
    assert bg._knn_edge_index(torch.zeros(1, 4), 3).shape == (2, 0)


def test_early_stopping_counts_down():
    from net_utils import EarlyStopping
    es = EarlyStopping(patience=2)
    m = torch.nn.Linear(2, 2)
    assert es(1.0, m) is False and es.counter == 2
    assert es(1.5, m) is False and es.counter == 1
    assert es(1.2, m) is True and es.get_best_model_state() is not None


def test_early_stopping_matches_reference_golden():
    """`net_utils.EarlyStopping` (reference `net_utils.py:130-158`): counter counts DOWN from patience, returns True at
    zero, keeps the best state -- against the reference's own object over a fixed loss sequence (golden)."""
    import net_utils as nu
    g = np.load(os.path.join(ROOT, "tests", "golden", "net_utils_loops.npz"), allow_pickle=False)
    model = torch.nn.Linear(2, 2)
    es = nu.EarlyStopping(patience=3, neptune_run=None)
    best_w = None
    for l, flag, counter, best in zip(g["es_losses"], g["es_flags"], g["es_counters"], g["es_best"]):
        with torch.no_grad():
            model.weight.add_(1.0)
        if l < es.best_loss:
            best_w = model.weight.detach().clone()
        assert bool(es(float(l), model)) == bool(flag)
        assert es.counter == int(counter) and es.best_loss == float(best)
    assert torch.equal(es.get_best_model_state()["weight"], best_w)


def test_derm_dataset_dict_contract(tmp_path):
    """`DermDataset.__getitem__` (reference `dataset.py:21-56`): the ten keys in the reference's order, dtypes and
    shapes, the 'no_mask' path, defaults for absent clinical columns, radiomics = zeros(102) unless a frame is given,
    and default collation into the dict batch `net_utils` consumes (`net_utils.py:11-19`)."""
    import pandas as pd
    from PIL import Image
    from dataset import ARTIFACT_COLS, DermDataset
    rng = np.random.RandomState(0)
    paths = []
    for i, (h, w) in enumerate([(40, 60), (50, 50)]):
        ip = str(tmp_path / f"img{i}.png")
        Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(ip)
        paths.append(ip)
    mask = np.zeros((40, 60), dtype=np.uint8)
    mask[10:20, 40:55] = 255                                   # lesion off-centre: the square crop follows it
    mp = str(tmp_path / "mask0.png")
    Image.fromarray(mask).save(mp)
    df = pd.DataFrame({"image_path": paths, "segmentation_path": [mp, "no_mask"], "dx": [3, 5],
                       "age_normalized": [0.25, 0.75], "sex_encoded": [1, 2], "loc_encoded": [7, 14],
                       **{c: [i % 2, (i + 1) % 2] for i, c in enumerate(ARTIFACT_COLS)}})
    ds = DermDataset(df, None)
    assert len(ds) == 2
    it = ds[0]
    assert list(it.keys()) == ["image", "mask", "radiomics", "age", "sex", "loc", "artifacts", "target", "image_path",
                               "segmentation_path"]
    assert it["image"].shape == (3, 40, 40) and it["image"].dtype == torch.float32 and float(it["image"].max()) <= 1.0
    assert it["mask"].shape == (1, 40, 40) and float(it["mask"].sum()) == 10 * 15       # the whole lesion is inside the crop
    assert it["radiomics"].shape == (102,) and it["radiomics"].dtype == torch.float32 and not it["radiomics"].any()
    assert it["age"].dtype == torch.float32 and abs(float(it["age"]) - 0.25) < 1e-7
    assert it["sex"].dtype == torch.long and int(it["sex"]) == 1 and int(it["loc"]) == 7
    assert it["artifacts"].dtype == torch.long and it["artifacts"].tolist() == [0, 1, 0, 1, 0, 1]
    assert it["target"].dtype == torch.long and int(it["target"]) == 3
    assert it["image_path"] == paths[0] and it["segmentation_path"] == mp
    it1 = ds[1]
    assert it1["image"].shape == (3, 50, 50) and not it1["mask"].any()                 # 'no_mask' -> empty mask
    # absent clinical / artifact columns fall back to the reference's defaults
    bare = DermDataset(df[["image_path", "segmentation_path", "dx"]], None)[1]
    assert float(bare["age"]) == 0.0 and int(bare["sex"]) == 0 and int(bare["loc"]) == 0 and not bare["artifacts"].any()
    # a radiomics frame (the north-star path) replaces the zero stub
    rad = pd.DataFrame(rng.randn(2, 12).astype(np.float32))
    assert torch.equal(DermDataset(df, rad)[1]["radiomics"], torch.as_tensor(rad.iloc[1].values))
    # default collation -> the dict batch net_utils moves to the device key by key
    ds2 = DermDataset(pd.concat([df.iloc[[1]]] * 3, ignore_index=True), None)
    batch = next(iter(torch.utils.data.DataLoader(ds2, batch_size=3)))
    assert batch["image"].shape == (3, 3, 50, 50) and batch["radiomics"].shape == (3, 102)
    assert batch["artifacts"].shape == (3, 6) and batch["target"].tolist() == [5, 5, 5] and len(batch["image_path"]) == 3


def test_vit_encoder_host_surface_and_oracle():
    """ViT-S/16 (BASELINE.json configs[4]) without a GPU: timm parameter names and shapes, frozen parameters, no CPU
    fallback; the fp32 oracle's fp16-emulating mode stays within fp16 noise of its fp32 mode."""
    import torch
    from isic_hip.lib import IsicHipError
    from isic_hip.vit import ViTSmallEncoder
    from oracle import vit as ov
    enc = ViTSmallEncoder(img_size=32)
    shapes = ov.vit_shapes(img=32)
    sd = enc.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    assert sum(v.numel() for v in ViTSmallEncoder().state_dict().values()) == 21_664_896      # timm vit_small_patch16_224 (22,050,664) minus head, cls token and its position
    assert all(not p.requires_grad for p in enc.parameters())
    with pytest.raises(IsicHipError):
        enc.run_tokens(torch.zeros(1, 3, 32, 32))
    with pytest.raises(IsicHipError):
        enc.train()
    with pytest.raises(ValueError):
        enc.run_tokens(torch.zeros(1, 3, 48, 48))
    p = ov.init_params(1, img=32)
    enc.load_state_dict(p)
    assert all(torch.equal(enc.state_dict()[k], p[k]) for k in p)
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(0))
    a, b = ov.forward_tokens(p, x), ov.forward_tokens(p, x, emulate_fp16=True)
    assert a.shape == (2, 4, 384) and float((a - b).abs().max()) < 2e-2 * float(a.abs().max())
    assert enc.flops_per_image() > 0
    # a timm-shaped checkpoint (class token + its position row + classifier head) loads with strict=False:
    # row 0 of pos_embed is the class token's position and is dropped, cls_token is ignored, head.* is reported
    full = ViTSmallEncoder()
    timm = {k: v.clone() for k, v in full.state_dict().items()}
    pos197 = torch.randn(1, 197, 384, generator=torch.Generator().manual_seed(1))
    timm["pos_embed"] = pos197
    timm["cls_token"] = torch.zeros(1, 1, 384)
    timm["head.weight"], timm["head.bias"] = torch.zeros(1000, 384), torch.zeros(1000)
    res = full.load_state_dict(timm, strict=False)
    assert not res.missing_keys and sorted(res.unexpected_keys) == ["head.bias", "head.weight"]
    assert torch.equal(full.state_dict()["pos_embed"], pos197[:, 1:, :])
    with pytest.raises(RuntimeError):
        full.load_state_dict({**timm, "pos_embed": torch.zeros(1, 50, 384)}, strict=False)


def test_committed_pmc_traffic_profile_matches_the_kernels():
    """`roofline.traffic` comes from profiles/r04_pmc_traffic.json (rocprofv3 PMC passes, tools/collect_traffic.sh).
    bench.py refuses a section stamped with other kernel sources, which silently nulled the field in BENCH_r02 -- so a
    kernel commit that invalidates the profile must fail HERE until the profile is re-collected."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    prof = json.load(open(bench.TRAFFIC_PROFILE))
    for section, prefixes in bench.TRAFFIC_SOURCES.items():
        assert section in prof, f"{section}: no PMC traffic section; run tools/collect_traffic.sh on the GPU box"
        assert prof[section]["kernel_source_hash"] == bench.kernel_source_hash(prefixes), (
            f"{section}: kernels {prefixes} changed since the PMC traffic profile was taken; re-run "
            f"SECTIONS={section} tools/collect_traffic.sh on the GPU box and commit profiles/r04_pmc_traffic.json")
        assert prof[section]["hbm_bytes_per_launch"] > 0
    # and bench.py's reader accepts it for the default workloads
    assert bench.pmc_traffic("mil", bags_per_step=64, patches=64, image_size=224) > 0
    assert bench.pmc_traffic("gnn", graphs_per_step=668, nodes=196, hidden=128, knn_k=8) > 0
    assert bench.pmc_traffic("vit", images_per_step=2048, image_size=224) > 0


def test_torch_library_ops_are_registered_with_fake_implementations():
    """isic_hip/torch_ops.py registers torch.ops.isic_hip.*: schemas exist, none claims to mutate or alias, and the fake
    (meta) implementations give the shapes / dtypes the kernels produce -- checked without a GPU under FakeTensorMode
    (the real launches are compared in tests/test_torch_ops_gpu.py)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from isic_hip import torch_ops  # noqa: F401
    O = torch.ops.isic_hip
    for name in ("linear", "linear_backward", "layer_norm", "layer_norm_backward", "spmm", "spmm_backward", "attn_pool",
                 "attn_pool_backward", "softmax_rows", "softmax_rows_backward", "cross_entropy", "gemm_f32"):
        schema = getattr(O, name).default._schema
        assert not schema.is_mutable, name
        assert all(a.alias_info is None for a in list(schema.arguments) + list(schema.returns)), name
    with FakeTensorMode():
        x, w, b = torch.empty(10, 32), torch.empty(8, 32), torch.empty(8)
        assert O.linear(x, w, b, 1, 0, 1.0, 0, 0).shape == (10, 8)
        dx, dw, db = O.linear_backward(torch.empty(10, 8), x, w, torch.empty(10, 8), 1, 1.0, True, True, False)
        assert dx.shape == x.shape and dw.shape == w.shape and db.numel() == 0
        y, mean, rstd = O.layer_norm(x, torch.empty(32), torch.empty(32), None, 1e-5, True, 0, 1.0, 0, 0)
        assert y.shape == x.shape and mean.shape == (10,) and rstd.shape == (10,)
        rp, c, v = torch.empty(11, dtype=torch.int32), torch.empty(40, dtype=torch.int32), torch.empty(40)
        assert O.spmm(rp, c, v, rp, c, v, x, None, 1.0).shape == x.shape
        z, att, t = O.attn_pool(torch.empty(50, 16), torch.empty(24, 16), torch.empty(24), torch.empty(3, 8), torch.empty(3),
                                torch.empty(6, dtype=torch.int64), 16, 3)
        assert z.shape == (5, 16) and att.shape == (50, 3) and t.shape == (50, 24)
        loss, d = O.cross_entropy(torch.empty(5, 4), torch.empty(5, dtype=torch.int64), 0)
        assert loss.shape == () and d.shape == (5, 4)
        assert O.gemm_f32(torch.empty(32, 10), torch.empty(8, 32), True, True, None, 0).shape == (10, 8)
    assert torch_ops.drop_args(None) == (0, 1.0, 0, 0)


def _scan_requested_registers(name, body):
    """Walk one kernel's gfx950 assembly with the in-order vmcnt model: every VMEM instruction joins a queue; `s_waitcnt
    vmcnt(N)` retires all but the youngest N; an inline-asm global load's destination registers are 'requested' until the
    load retires, and no other instruction may read or write them meanwhile.  A backward branch replays its loop body once
    with the state at the branch (what the second iteration sees).  -> number of asm requests seen."""
    import re
    lines = [l.strip() for l in body.split("\n")]
    labels = {l[:-1].split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    queue, requests, replayed = [], 0, set()

    def regs_of(arg):
        out = set()
        for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", arg):
            out |= {int(m.group(3))} if m.group(3) else set(range(int(m.group(1)), int(m.group(2)) + 1))
        return out

    def run(lo, hi):
        nonlocal requests
        in_asm, i = False, lo
        while i < hi:
            t = lines[i]
            i += 1
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            op, _, rest = t.partition(" ")
            rest = rest.split(";")[0]
            m = re.search(r"vmcnt\((\d+)\)", t) if op == "s_waitcnt" else None
            if m:
                n = int(m.group(1))
                del queue[:max(0, len(queue) - n)]
                continue
            pending = set().union(*[q for q in queue if q]) if queue else set()
            parts = [x.strip() for x in rest.split(",")]
            touched = regs_of(rest)
            if op.startswith(("global_load", "global_store", "buffer_", "flat_", "scratch_")):
                assert not (touched & pending), f"{name}: '{t}' touches a requested register before its wait"
                if in_asm and op.startswith("global_load"):
                    queue.append(regs_of(parts[0]))
                    requests += 1
                else:
                    queue.append(None)
                continue
            assert not (touched & pending), f"{name}: '{t}' touches a requested register before its wait"
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = rest.strip()
                if tgt in labels and labels[tgt] < i and (tgt, i) not in replayed:
                    replayed.add((tgt, i))
                    run(labels[tgt], i - 1)
    run(0, len(lines))
    return requests


def _gfx950_kernels(src, pattern):
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", src, "-o", out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read()
    return re.findall(r"^(_ZN\S*" + pattern + r"[^\s:]*):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)


def test_hand_counted_gemm_loads_are_never_touched_before_their_wait():
    """gemm_f32r.hip (row panel) and gemm_f32t.hip (A^T B) request operands with inline-asm global loads that hipcc does not
    track -- its own s_waitcnt insertion drained the pipeline right after each request -- and count vmcnt by hand.  The
    price: any compiler-made copy of a requested register between the request and the wait (a phi copy at a join or at a loop's
    end, the set-up of a tied asm operand) reads a register the load has not written yet.  All three happened while the
    kernels were written; one showed up as a test failing one run in four.  This compiles both files to gfx950 assembly and
    checks every instantiation with the in-order vmcnt model."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "multimodal-isic_amd", "csrc")
    rp = _gfx950_kernels(os.path.join(csrc, "gemm_f32r.hip"), "gemm_rowpanel_kernel")
    assert len(rp) >= 16, "expected the KC = 1..8 x {plain, addend} instantiations"
    tn = _gfx950_kernels(os.path.join(csrc, "gemm_f32t.hip"), "gemm_tn_skinny_kernel")
    assert len(tn) == 1
    for name, body in rp + tn:
        assert _scan_requested_registers(name, body) > 0, name
    # the checker itself: a copy in front of the wait, and a loop that carries a requested register over its back-edge
    bad = ";;#ASMSTART\nglobal_load_dwordx4 v[10:13], v[2:3], off\n;;#ASMEND\nv_mov_b32_e32 v20, v11\ns_waitcnt vmcnt(0)\n"
    with pytest.raises(AssertionError):
        _scan_requested_registers("bad", bad)
    loop = (".LBB0_1:\nv_mov_b64_e32 v[20:21], v[10:11]\n;;#ASMSTART\nglobal_load_dwordx4 v[10:13], v[2:3], off\n;;#ASMEND\n"
            "s_cbranch_scc1 .LBB0_1\ns_waitcnt vmcnt(0)\n")
    with pytest.raises(AssertionError):
        _scan_requested_registers("loop", loop)
    ok = (";;#ASMSTART\nglobal_load_dwordx4 v[10:13], v[2:3], off\n;;#ASMEND\nglobal_store_dwordx4 v[4:5], v[30:33], off\n"
          ";;#ASMSTART\ns_waitcnt vmcnt(1)\n;;#ASMEND\nv_mov_b32_e32 v20, v11\n")
    assert _scan_requested_registers("ok", ok) == 1
