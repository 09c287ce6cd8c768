"""Formula-generated tensors (oracle / test infrastructure).

Golden fixtures are produced from closed-form inputs and weights so that no
weight files have to be committed: both ``oracle/gen_golden.py`` (which runs the
reference) and the tests (which run the oracle and the HIP path) rebuild the
very same tensors from these formulas.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch


def ftensor(shape, scale=1.0, freq=0.37, phase=0.0, dtype=torch.float32):
    """``t.flat[i] = scale * (sin(freq*i + phase) + 0.25*cos(0.013*i + 2*phase))``.

    Computed in float64 and rounded once to ``dtype`` so every consumer sees
    bit-identical values.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    v = scale * (np.sin(freq * i + phase) + 0.25 * np.cos(0.013 * i + 2.0 * phase))
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def formula_state_dict(shapes: "OrderedDict[str, tuple]", gain=1.0, dtype=torch.float32):
    """Deterministic state_dict for a module whose parameter shapes are ``shapes``.

    Weights of fan-in ``f`` get amplitude ``gain/sqrt(f)`` (activations stay
    O(1)); LayerNorm / BatchNorm weights sit around 1; biases are small.
    """
    out = OrderedDict()
    for idx, (name, shape) in enumerate(shapes.items()):
        phase = 0.61 * (idx + 1)
        shape = tuple(shape)
        leaf = name.rsplit(".", 1)[-1]
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            t = ftensor(shape, gain / math.sqrt(fan_in) * 1.6, 0.37 + 0.01 * idx, phase, dtype)
        elif leaf == "weight":  # 1-D weight = a norm layer's gamma
            t = 1.0 + ftensor(shape, 0.1, 0.53, phase, dtype)
        else:
            t = ftensor(shape, 0.05, 0.71, phase, dtype)
        out[name] = t
    return out


def formula_input(n, d, phase=0.5, scale=1.0):
    """Bag / node features ``x[n, d]``."""
    return ftensor((n, d), scale, 0.11, phase)


def shapes_of(module) -> "OrderedDict[str, tuple]":
    return OrderedDict((k, tuple(v.shape)) for k, v in module.state_dict().items())


def gapped_points(n, d, seed=0):
    """Points whose pairwise distances are all separated by a clear gap, so
    that k-NN sets are insensitive to fp32 summation order
    (reference `03_build_graphs.py:46-50` computes ||x||^2+||y||^2-2xy in fp32)."""
    rng = np.random.RandomState(seed)
    base = rng.randn(n, d).astype(np.float64)
    # geometric radial scaling -> distances from any node differ by >= ~2 %
    r = (1.03 ** np.arange(n))[:, None]
    x = base / np.linalg.norm(base, axis=1, keepdims=True) * r
    return torch.from_numpy(x).to(torch.float32)
