"""Shared helpers for the tests (fixture decoding, tolerances)."""
import os
from collections import OrderedDict

import numpy as np
import torch

from oracle import formula

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def shapes_from_blob(blob):
    out = OrderedDict()
    for s in blob.tolist():
        k, dims = s.split("|")
        out[k] = tuple(int(v) for v in dims.split(",")) if dims else ()
    return out


def formula_params(g):
    return formula.formula_state_dict(shapes_from_blob(g["names"]))


def check_grad(g, key, t, rtol=2e-4, atol=1e-6):
    """Compare tensor ``t`` against golden gradient ``key`` (full or sampled)."""
    a = t.detach().cpu().double().numpy().reshape(-1)
    if f"grad.{key}" in g.files:
        ref = g[f"grad.{key}"].reshape(-1).astype(np.float64)
        scale = max(np.abs(ref).max(), 1e-30)
        np.testing.assert_allclose(a, ref, rtol=rtol, atol=atol + rtol * scale)
        return
    stride = int(g[f"grad.{key}#stride"])
    ref = g[f"grad.{key}#sample"].astype(np.float64)
    scale = max(np.abs(ref).max(), 1e-30)
    np.testing.assert_allclose(a[::stride][: ref.size], ref, rtol=rtol, atol=atol + rtol * scale)
    l2 = float(g[f"grad.{key}#l2"])
    assert abs(np.sqrt((a * a).sum()) - l2) <= rtol * max(l2, 1e-30) + atol
    assert a.size == int(g[f"grad.{key}#numel"])


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    scale = max(np.abs(b).max(), 1e-30) if b.size else 1.0
    err = np.abs(a - b).max() if b.size else 0.0
    assert err <= atol + rtol * scale, f"{what}: max|diff|={err:.3e} > {atol + rtol * scale:.3e} (scale {scale:.3e})"
