"""ViT-S/16 patch encoder on torch-CPU in fp32 (oracle / test infrastructure).

PARITY UNPINNED against the reference: its frozen encoder is an un-vendored ConvMAE conv-ViT
(`save_latent.py:17-18,42-60`, `.gitignore:6`) with no code or weights in the tree; `BASELINE.json`
configs[4] names "ViT-S patch encoder, fp16" instead.  This file is the build's own definition of that
encoder -- Dosovitskiy et al. 2020 / timm `vit_small_patch16_224` without the class token (the reference
consumes the 196 patch tokens only, `save_latent.py:56-60,77`): 16x16 patch projection, learned position
embedding, 12 pre-norm blocks (LayerNorm eps 1e-6, 6-head attention of width 64, erf-GELU MLP of ratio 4),
final LayerNorm -- and it is what the HIP fp16 path (`isic_hip/vit.py`) is checked against.

``emulate_fp16=True`` rounds to fp16 at exactly the points where the HIP path stores fp16 (patch rows,
weights, every Linear / LayerNorm / attention output, the residual stream), with fp32 arithmetic in
between, so that the comparison isolates summation order.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

DIM, DEPTH, HEADS, MLP, PATCH, LN_EPS = 384, 12, 6, 1536, 16, 1e-6


def vit_shapes(in_ch=3, img=224, dim=DIM, depth=DEPTH, mlp=MLP, patch=PATCH):
    s = OrderedDict()
    s["patch_embed.proj.weight"] = (dim, in_ch, patch, patch)
    s["patch_embed.proj.bias"] = (dim,)
    s["pos_embed"] = (1, (img // patch) ** 2, dim)
    for i in range(depth):
        p = f"blocks.{i}"
        s[f"{p}.norm1.weight"] = (dim,); s[f"{p}.norm1.bias"] = (dim,)
        s[f"{p}.attn.qkv.weight"] = (3 * dim, dim); s[f"{p}.attn.qkv.bias"] = (3 * dim,)
        s[f"{p}.attn.proj.weight"] = (dim, dim); s[f"{p}.attn.proj.bias"] = (dim,)
        s[f"{p}.norm2.weight"] = (dim,); s[f"{p}.norm2.bias"] = (dim,)
        s[f"{p}.mlp.fc1.weight"] = (mlp, dim); s[f"{p}.mlp.fc1.bias"] = (mlp,)
        s[f"{p}.mlp.fc2.weight"] = (dim, mlp); s[f"{p}.mlp.fc2.bias"] = (dim,)
    s["norm.weight"] = (dim,); s["norm.bias"] = (dim,)
    return s


def init_params(seed=0, **kw):
    """Deterministic initialisation shared by the oracle and the HIP module: trunc-normal-like N(0, 0.02) weights,
    slightly perturbed LayerNorm affine and biases (so that a dropped bias or gamma shows up in a test)."""
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for k, shp in vit_shapes(**kw).items():
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k == "norm.weight":
            p[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            p[k] = 0.02 * torch.randn(shp, generator=g)
        elif k == "patch_embed.proj.weight":
            fan = shp[1] * shp[2] * shp[3]
            p[k] = torch.randn(shp, generator=g) / math.sqrt(fan)
        elif k == "pos_embed":
            p[k] = 0.02 * torch.randn(shp, generator=g)
        else:
            p[k] = torch.randn(shp, generator=g) / math.sqrt(shp[1])
    return p


def _r(x, on):
    return x.half().float() if on else x


def forward_tokens(p, images, emulate_fp16=False, heads=HEADS, depth=None, return_blocks=False):
    """images[N,3,H,W] fp32 -> tokens[N, (H/16)*(W/16), 384] fp32 (the final LayerNorm is returned unrounded)."""
    e = emulate_fp16
    w = {k: (_r(v, e) if (v.dim() > 1 and k != "pos_embed") else v) for k, v in p.items()}   # fp16 weights; fp32 bias / LN
    patch = p["patch_embed.proj.weight"].shape[-1]
    dim = p["patch_embed.proj.weight"].shape[0]
    N = images.shape[0]
    x = F.conv2d(_r(images, e), w["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=patch)
    x = x.flatten(2).transpose(1, 2)                                        # [N, T, D]
    x = _r(x + _r(p["pos_embed"], e), e)
    T = x.shape[1]
    hd = dim // heads
    outs = []
    nblocks = depth if depth is not None else sum(1 for k in p if k.endswith(".norm1.weight"))
    for i in range(nblocks):
        b = f"blocks.{i}"
        h = _r(F.layer_norm(x, (dim,), p[f"{b}.norm1.weight"], p[f"{b}.norm1.bias"], LN_EPS), e)
        qkv = _r(F.linear(h, w[f"{b}.attn.qkv.weight"], p[f"{b}.attn.qkv.bias"]), e)
        q, k, v = qkv.view(N, T, 3, heads, hd).permute(2, 0, 3, 1, 4)        # [N, heads, T, hd]
        a = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
        o = (_r(a, e) @ v) if e else (a @ v)
        o = _r(o.transpose(1, 2).reshape(N, T, dim), e)
        x = _r(x + F.linear(o, w[f"{b}.attn.proj.weight"], p[f"{b}.attn.proj.bias"]), e)
        h = _r(F.layer_norm(x, (dim,), p[f"{b}.norm2.weight"], p[f"{b}.norm2.bias"], LN_EPS), e)
        h = _r(F.gelu(F.linear(h, w[f"{b}.mlp.fc1.weight"], p[f"{b}.mlp.fc1.bias"])), e)
        x = _r(x + F.linear(h, w[f"{b}.mlp.fc2.weight"], p[f"{b}.mlp.fc2.bias"]), e)
        outs.append(x)
    y = F.layer_norm(x, (dim,), p["norm.weight"], p["norm.bias"], LN_EPS)
    return (y, outs) if return_blocks else y
