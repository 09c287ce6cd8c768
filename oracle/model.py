"""The composed bag model (encoder -> MIL teacher head -> radiomic fusion) on torch-CPU
(oracle / test infrastructure).

PARITY UNPINNED as a whole: the reference contains the pieces -- the MIL teacher
head (`utils_g_mil.py:38-105`, pinned), the ``Linear->LayerNorm->ReLU->Dropout``
modality MLPs and the intermediate fusion (`model.py:63-83,206-216`, pinned) -- but
never composes bags of images with radiomics, and has no ResNet-18.
This file is the build's definition of ``model.MultiModalMILNet`` and is the CPU
baseline timed by ``bench.py`` (per-bag optimizer steps as
`01_train_mil_teacher.py:237-246`).
"""
from __future__ import annotations

import time

import torch
import torch.nn.functional as F

from . import fusion, mil, resnet


def sub(p, prefix):
    n = len(prefix) + 1
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix + ".")}


def milnet_forward(p, image, radiomics, offsets, emulate_bf16=False, layers=resnet.LAYERS, fusion_strategy="concat",
                   drop=None, mil_dropout=0.0, running=None, training=True, acc64=False):
    """image[T,3,H,W], radiomics[B,R], offsets[B+1] -> dict like model.MultiModalMILNet.

    BatchNorm uses batch statistics (``training=True``; ``running`` = encoder-relative buffer dict updated in place when
    given) or the running statistics (``training=False``).  ``drop=None``: no dropout;
    ``drop=dict(seed, step)``: the product's counter-based train-mode dropout -- sites
    image_proj (p .3/.2, streams step*1024 + 0/1), radiomics_mlp (.4/.3, +2/3), fusion_mlp (.4, +8)
    under ``seed``; the MIL head's own site (p = mil_dropout, stream step*1024) under ``seed + 1``."""
    feats = resnet.resnet18_features(sub(p, "encoder"), image, emulate_bf16=emulate_bf16, layers=layers,
                                     running=running, training=training, acc64=acc64)
    mdrop = None
    fdrop = None
    if drop is not None:
        fdrop = {"seed": drop["seed"], "stream_base": drop["step"] * 1024}
        if mil_dropout > 0:
            mdrop = {"p": mil_dropout, "seed": drop["seed"] + 1, "stream": drop["step"] * 1024}
    out = mil.teacher_forward_batched(sub(p, "mil"), feats, offsets, drop=mdrop)
    H = p["mil.feature_extractor.0.weight"].shape[0]
    pooled = []
    for b in range(len(offsets) - 1):
        lo, hi = int(offsets[b]), int(offsets[b + 1])
        pooled.append((out["attention"][lo:hi, None] * out["hidden"][lo:hi]).sum(dim=0) if hi > lo
                      else torch.zeros(H))
    z = torch.stack(pooled)
    img = fusion.mlp_ln_relu(p, "image_proj", z, fdrop, (0.3, 0.2), 0)
    rad = fusion.mlp_ln_relu(p, "radiomics_mlp", radiomics, fdrop, (0.4, 0.3), 2)
    if fusion_strategy == "concat":
        fused = torch.cat([img, rad], dim=1)
    else:
        fused, _ = fusion.attention_fusion(p, [img, rad])
    out["logits"] = fusion.fusion_mlp(p, fused, fdrop, 8)
    out["features"] = feats
    return out


def milnet_loss(out, y, aux_weight=1.0):
    l = F.cross_entropy(out["logits"], y.long())
    if aux_weight:
        l = l + aux_weight * F.cross_entropy(out["bag_logits"], y.long())
    return l


def time_per_bag_train_loop(p, make_bag, n_bags, warmup, lr=2.2e-4, weight_decay=8.6e-4, layers=resnet.LAYERS,
                            budget_s=25.0):
    """CPU baseline: the reference training loop shape (`01_train_mil_teacher.py:235-246`: one bag
    per optimizer step, fp32, torch CPU, AdamW) over the composed model.  ``make_bag(i)`` returns
    (image[K,3,H,W], radiomics[1,R], label).  Stops after ``n_bags`` timed steps or ``budget_s``
    seconds, whichever comes first.  Returns (bags_per_second, timed_bags)."""
    q = {k: torch.nn.Parameter(v.detach().clone().float()) for k, v in p.items()}
    opt = torch.optim.AdamW(list(q.values()), lr=lr, weight_decay=weight_decay)

    def step(i):
        img, rad, y = make_bag(i)
        opt.zero_grad()
        out = milnet_forward(q, img, rad, [0, img.shape[0]], layers=layers)
        milnet_loss(out, torch.as_tensor([y])).backward()
        opt.step()

    for i in range(warmup):
        step(i)
    t0 = time.perf_counter()
    done = 0
    for i in range(n_bags):
        step(warmup + i)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return done / (time.perf_counter() - t0), done
