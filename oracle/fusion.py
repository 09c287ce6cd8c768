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


def attention_fusion_late(p, logits, prefix="attention"):
    """`model.py:25-40` (AttentionFusion_Late): scores = Linear(M*C,128)->ReLU->Linear(128,M) on the concatenated
    per-modality logits; softmax over M; weighted sum of the logits."""
    cat = torch.cat(logits, dim=1)
    s = F.linear(F.relu(F.linear(cat, p[f"{prefix}.attention_net.0.weight"], p[f"{prefix}.attention_net.0.bias"])),
                 p[f"{prefix}.attention_net.2.weight"], p[f"{prefix}.attention_net.2.bias"])
    w = torch.softmax(s, dim=1).unsqueeze(2)
    return (torch.stack(logits, dim=1) * w).sum(dim=1)


def late_fusion(p, feats, modality, strategy):
    """`model.py:155-164` (one Linear(128,C) head per modality) + the late branches `model.py:216-227`:
    "concat" is a plain sum of the per-modality logits, "weighted" a softmax(weights)-weighted sum,
    "attention" AttentionFusion_Late."""
    logits = [F.linear(f, p[f"modality_heads.{m}.weight"], p[f"modality_heads.{m}.bias"])
              for m, f in zip(modality, feats)]
    if strategy == "concat":
        return torch.stack(logits, dim=1).sum(dim=1)
    if strategy == "weighted":
        nw = torch.softmax(p["weights"], dim=0)
        return torch.stack([w * z for w, z in zip(nw, logits)], dim=0).sum(dim=0)
    if strategy == "attention":
        return attention_fusion_late(p, logits)
    raise ValueError(strategy)
