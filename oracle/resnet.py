"""ResNet-18 patch encoder on torch-CPU (oracle / test infrastructure).

PARITY UNPINNED against the reference: the reference's patch encoder is an
un-vendored ConvMAE (`save_latent.py:17-18,42-60`, `.gitignore:6`) with no
weights in the tree; `BASELINE.json` configs[1] names ResNet-18 instead.  This
file is the build's own definition of that encoder (He et al. 2015 basic-block
ResNet-18, torchvision parameter names, no fc: global-average-pooled 512-d
features), and it is what the HIP implicit-GEMM convolutions are checked
against (SURVEY.md §8d layer table).

``emulate_bf16=True`` rounds at exactly the points where the HIP path stores
bf16 (conv inputs/weights/outputs, BN+ReLU outputs, activation gradients), with
fp32 accumulation in between, so that the comparison isolates summation order.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

LAYERS = ((64, 1), (128, 2), (256, 2), (512, 2))  # (planes, stride of first block)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def resnet18_shapes(in_ch=3, layers=LAYERS):
    s = OrderedDict()

    def bn(prefix, c):
        s[f"{prefix}.weight"] = (c,)
        s[f"{prefix}.bias"] = (c,)

    s["conv1.weight"] = (64, in_ch, 7, 7)
    bn("bn1", 64)
    inp = 64
    for li, (planes, stride) in enumerate(layers, start=1):
        for b in range(2):
            pre = f"layer{li}.{b}"
            st = stride if b == 0 else 1
            s[f"{pre}.conv1.weight"] = (planes, inp, 3, 3)
            bn(f"{pre}.bn1", planes)
            s[f"{pre}.conv2.weight"] = (planes, planes, 3, 3)
            bn(f"{pre}.bn2", planes)
            if st != 1 or inp != planes:
                s[f"{pre}.downsample.0.weight"] = (planes, inp, 1, 1)
                bn(f"{pre}.downsample.1", planes)
            inp = planes
    return s


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _r(x, on):
    return _RoundBF16.apply(x) if on else x


def _bn_train(x, w, b, stats, name, running=None):
    # batch statistics over (N,H,W), biased variance for normalisation; ``running``: torch BatchNorm2d's running
    # estimates ({name}.running_mean / .running_var, momentum 0.1, UNBIASED variance) updated in place
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    if stats is not None:
        stats[name] = (mean.detach(), var.detach())
    if running is not None:
        n = x.numel() // x.shape[1]
        unb = var.detach() * (n / (n - 1.0)) if n > 1 else var.detach()
        running[f"{name}.running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
        running[f"{name}.running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unb)
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * w)[None, :, None, None] + b[None, :, None, None]


def _bn_eval(x, w, b, running, name):
    # y = x * scale + shift with scale = gamma * rsqrt(running_var + eps): the form isic_bn_eval_affine folds to
    scale = w * torch.rsqrt(running[f"{name}.running_var"] + BN_EPS)
    shift = b - running[f"{name}.running_mean"] * scale
    return x * scale[None, :, None, None] + shift[None, :, None, None]


def fresh_running(p):
    """Initial BatchNorm buffers (mean 0, var 1) for every BatchNorm of the parameter dict ``p``."""
    out = {}
    for k, v in p.items():
        if v.dim() == 1 and k.endswith(".weight"):
            name = k[:-len(".weight")]
            out[f"{name}.running_mean"] = torch.zeros_like(v)
            out[f"{name}.running_var"] = torch.ones_like(v)
    return out


def resnet18_features(p, x, emulate_bf16=False, stats=None, taps=None, layers=LAYERS, running=None, training=True, acc64=False):
    """x[N,3,H,W] fp32 -> features[N,512] fp32.  ``training=True``: batch-statistics BatchNorm (and the running
    estimates in ``running`` are updated when given); ``training=False``: normalise with ``running``.  ``layers``
    shortens the network for well-conditioned tests (default: the four ResNet-18 stages).
    ``acc64``: every convolution (forward, data and weight gradient) accumulates in float64 and is rounded to float32 once --
    the SAME arithmetic in another summation order.  With ``emulate_bf16`` this is the yardstick for what a different
    summation order (the MFMA's, say) does to a bf16 pipeline: the float32 values in front of each bf16 rounding move by
    ~1e-7 relative, a few of the millions of activations / gradients round the other way, and training amplifies that."""
    e = emulate_bf16
    if not training and running is None:
        raise ValueError("eval-mode BatchNorm needs the running statistics")

    def conv(t, name, stride, pad):
        w = p[name]
        if e:
            w = _RoundBF16.apply(w)
        if acc64:
            return _r(F.conv2d(t.double(), w.double(), None, stride, pad).float(), e)
        return _r(F.conv2d(t, w, None, stride, pad), e)

    def cbr(t, cname, bname, stride, pad, relu=True, residual=None):
        c = conv(t, cname, stride, pad)
        if training:
            y = _bn_train(c, p[f"{bname}.weight"], p[f"{bname}.bias"], stats, bname, running)
        else:
            y = _bn_eval(c, p[f"{bname}.weight"], p[f"{bname}.bias"], running, bname)
        if residual is not None:
            y = y + residual
        if relu:
            y = F.relu(y)
        return _r(y, e)

    t = _r(x, e)
    t = cbr(t, "conv1.weight", "bn1", 2, 3)
    if taps is not None:
        taps["stem"] = t
    t = F.max_pool2d(t, 3, 2, 1)
    if taps is not None:
        taps["pool"] = t
    inp = 64
    for li, (planes, stride) in enumerate(layers, start=1):
        for b in range(2):
            pre = f"layer{li}.{b}"
            st = stride if b == 0 else 1
            idt = t
            if f"{pre}.downsample.0.weight" in p:
                idt = cbr(t, f"{pre}.downsample.0.weight", f"{pre}.downsample.1", st, 0, relu=False)
            u = cbr(t, f"{pre}.conv1.weight", f"{pre}.bn1", st, 1)
            t = cbr(u, f"{pre}.conv2.weight", f"{pre}.bn2", 1, 1, relu=True, residual=idt)
            inp = planes
            if taps is not None:
                taps[pre] = t
    return t.mean(dim=(2, 3))
