"""Attention-MIL heads, restated on torch-CPU (oracle / test infrastructure).

Follows the reference classes
  * ``AttentionMIL_teacher``  -- `utils_g_mil.py:38-105`  (class-space pooling)
  * ``AttentionMIL``          -- `utils_g_mil.py:15-36`   (feature-space pooling)
and the per-bag train step of `01_train_mil_teacher.py:237-246`.

Parameters come in as a dict keyed like the reference ``state_dict``
(SURVEY.md §8 a1):
  feature_extractor.0.{weight[H,D],bias[H]}  attention.0.{weight[A,H],bias[A]}
  attention.2.{weight[1,A],bias[1]}          patch_classifier|classifier.{weight[C,H],bias[C]}
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import philox


def dropout(x, p, seed, stream, training=True, elem_offset=0):
    """Counter-based dropout (definition in ``oracle/philox.py``).  ``elem_offset``: index of ``x.flat[0]`` in the stream
    (a bag / graph taken out of a batch keeps the words of its position in the batched tensor)."""
    if (not training) or p <= 0.0:
        return x
    n = x.numel()
    keep = torch.from_numpy(philox.dropout_keep(elem_offset + n, p, seed, stream)[elem_offset:elem_offset + n].copy()).view(x.shape)
    scale = torch.tensor(float(philox.dropout_scale(p)), dtype=x.dtype)
    return torch.where(keep, x * scale, torch.zeros((), dtype=x.dtype))


def _hidden(p, x, drop=None, elem_offset=0):
    # utils_g_mil.py:49-53,69  Linear -> ReLU -> Dropout
    h = F.relu(F.linear(x, p["feature_extractor.0.weight"], p["feature_extractor.0.bias"]))
    if drop is not None and drop["p"] > 0.0:
        n = h.numel()
        keep_all = philox.dropout_keep(elem_offset + n, drop["p"], drop["seed"], drop["stream"])
        keep = torch.from_numpy(keep_all[elem_offset:elem_offset + n].copy()).view(h.shape)
        scale = torch.tensor(float(philox.dropout_scale(drop["p"])), dtype=h.dtype)
        h = torch.where(keep, h * scale, torch.zeros((), dtype=h.dtype))
    return h


def _attention_logits(p, h):
    # utils_g_mil.py:55-59,72  Linear -> Tanh -> Linear(A,1)
    t = torch.tanh(F.linear(h, p["attention.0.weight"], p["attention.0.bias"]))
    return F.linear(t, p["attention.2.weight"], p["attention.2.bias"])  # [N,1]


def teacher_forward(p, x, drop=None, elem_offset=0):
    """`utils_g_mil.py:66-105`.  x[N,D] -> dict of 5 tensors."""
    h = _hidden(p, x, drop, elem_offset)
    a = torch.softmax(_attention_logits(p, h), dim=0)                      # :73-76
    patch_logits = F.linear(h, p["patch_classifier.weight"], p["patch_classifier.bias"])  # :79
    bag_logits = torch.sum(a * patch_logits, dim=0)                        # :83-86
    return {
        "bag_logits": bag_logits,
        "bag_probs": torch.softmax(bag_logits, dim=0),                     # :89-92
        "attention": a.squeeze(-1),
        "patch_logits": patch_logits,
        "patch_probs": torch.softmax(patch_logits, dim=1),                 # :94-97
        "hidden": h,
    }


def attention_mil_forward(p, x, drop=None, elem_offset=0):
    """`utils_g_mil.py:30-36`.  x[N,D] -> (probs[C], a[N,1])."""
    h = _hidden(p, x, drop, elem_offset)
    a = torch.softmax(_attention_logits(p, h), dim=0)
    z = torch.sum(a * h, dim=0)
    logits = F.linear(z, p["classifier.weight"], p["classifier.bias"])
    return torch.softmax(logits, dim=0), a, logits, z


def teacher_forward_batched(p, x, offsets, drop=None):
    """The build's batched form: ``x[sum K, D]`` + CSR ``offsets[B+1]``; every bag is
    pooled independently exactly as one reference call would (dropout element
    index = position in the concatenated ``h``)."""
    outs = []
    H = p["feature_extractor.0.weight"].shape[0]
    for b in range(len(offsets) - 1):
        lo, hi = int(offsets[b]), int(offsets[b + 1])
        outs.append(teacher_forward(p, x[lo:hi], drop, elem_offset=lo * H))
    return {
        "bag_logits": torch.stack([o["bag_logits"] for o in outs]),
        "bag_probs": torch.stack([o["bag_probs"] for o in outs]),
        "attention": torch.cat([o["attention"] for o in outs]),
        "patch_logits": torch.cat([o["patch_logits"] for o in outs]),
        "patch_probs": torch.cat([o["patch_probs"] for o in outs]),
        "hidden": torch.cat([o["hidden"] for o in outs]),
    }


def bag_loss(bag_logits, y):
    """`01_train_mil_teacher.py:143,244`: CrossEntropyLoss()(logits[None], y)."""
    return F.cross_entropy(bag_logits.unsqueeze(0), y.view(1).long())


def batched_loss(bag_logits, y):
    """Mean over the bags of a step of the per-bag reference loss (SURVEY.md §7
    "Batch-vs-per-bag semantics")."""
    return F.cross_entropy(bag_logits, y.long())


def teacher_loss_and_grads(p, x, y, offsets=None):
    """Loss + gradient of every parameter (and of x).  Single bag when
    ``offsets`` is None."""
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    xx = x.detach().clone().requires_grad_(True)
    if offsets is None:
        out = teacher_forward(q, xx)
        loss = bag_loss(out["bag_logits"], y)
    else:
        out = teacher_forward_batched(q, xx, offsets)
        loss = batched_loss(out["bag_logits"], y)
    loss.backward()
    grads = {k: v.grad.detach() for k, v in q.items()}
    grads["x"] = xx.grad.detach()
    return loss.detach(), {k: v.detach() for k, v in out.items()}, grads


def per_bag_train_loop(p, bags, labels, steps, lr, weight_decay, optimizer="adamw"):
    """The reference hot loop, `01_train_mil_teacher.py:235-246`: one optimizer
    step per bag, torch AdamW/Adam (`01:217-224`), eval-mode dropout off here
    (callers pass drop via teacher_forward when they need it).  Returns the
    parameter dict after each step."""
    q = {k: torch.nn.Parameter(v.detach().clone()) for k, v in p.items()}
    cls = torch.optim.AdamW if optimizer == "adamw" else torch.optim.Adam
    opt = cls(list(q.values()), lr=lr, weight_decay=weight_decay)
    history = []
    for s in range(steps):
        x, y = bags[s % len(bags)], labels[s % len(bags)]
        opt.zero_grad()
        out = teacher_forward(q, x)
        loss = bag_loss(out["bag_logits"], torch.as_tensor(y))
        loss.backward()
        opt.step()
        history.append(({k: v.detach().clone() for k, v in q.items()}, float(loss.detach())))
    return history
