"""Radiomic MLP + modality fusion, restated on torch-CPU (oracle / test infrastructure).

Follows `model.py:6-40` (AttentionFusion, AttentionFusion_Late), `model.py:63-83`
(image_proj / radiomics_mlp: Linear->LayerNorm->ReLU->Dropout x2) and the fusion
branches `model.py:206-227`.  Eval-mode (dropout off) unless ``drop`` given.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import mil as _mil


def mlp_ln_relu(p, prefix, x, drop=None, p_drop=(0.4, 0.3), stream=0):
    """`model.py:74-83`: Linear(R,256) LN ReLU Drop(.4) Linear(256,128) LN ReLU Drop(.3).
    Parameter names ``{prefix}.0/.1/.4/.5``."""
    h = F.linear(x, p[f"{prefix}.0.weight"], p[f"{prefix}.0.bias"])
    h = F.relu(F.layer_norm(h, (h.shape[-1],), p[f"{prefix}.1.weight"], p[f"{prefix}.1.bias"]))
    if drop is not None:
        h = _mil.dropout(h, p_drop[0], drop["seed"], drop["stream_base"] + stream)
    h = F.linear(h, p[f"{prefix}.4.weight"], p[f"{prefix}.4.bias"])
    h = F.relu(F.layer_norm(h, (h.shape[-1],), p[f"{prefix}.5.weight"], p[f"{prefix}.5.bias"]))
    if drop is not None:
        h = _mil.dropout(h, p_drop[1], drop["seed"], drop["stream_base"] + stream + 1)
    return h


def attention_fusion(p, feats, prefix="attention"):
    """`model.py:15-23`: stack [B,M,D]; scores = Linear(D,128)->Tanh->Linear(128,1);
    softmax over M; weighted sum."""
    st = torch.stack(feats, dim=1)
    s = F.linear(torch.tanh(F.linear(st, p[f"{prefix}.attn.0.weight"], p[f"{prefix}.attn.0.bias"])),
                 p[f"{prefix}.attn.2.weight"], p[f"{prefix}.attn.2.bias"]).squeeze(-1)
    w = torch.softmax(s, dim=1).unsqueeze(-1)
    return (st * w).sum(dim=1), w.squeeze(-1)


def fusion_mlp(p, fused, drop=None, stream=8):
    """`model.py:129-143`: Linear(.,256)->ReLU->Dropout(.4)->Linear(256,C)."""
    h = F.relu(F.linear(fused, p["fusion_mlp.0.weight"], p["fusion_mlp.0.bias"]))
    if drop is not None:
        h = _mil.dropout(h, 0.4, drop["seed"], drop["stream_base"] + stream)
    return F.linear(h, p["fusion_mlp.3.weight"], p["fusion_mlp.3.bias"])


def intermediate_fusion(p, feats, strategy, drop=None):
    """`model.py:206-216`."""
    if strategy == "concat":
        fused = torch.cat(feats, dim=1)
    elif strategy == "weighted":
        nw = torch.softmax(p["weights"], dim=0)
        fused = torch.cat([w * f for w, f in zip(nw, feats)], dim=1)
    elif strategy == "attention":
        fused, _ = attention_fusion(p, feats)
    else:
        raise ValueError(strategy)
    return fusion_mlp(p, fused, drop)
