"""Flat-buffer Adam / AdamW on the MI355X (one HIP launch per step).

Same update rule and order of operations as ``torch.optim.AdamW`` / ``Adam``
(reference `01_train_mil_teacher.py:217-224`, `05_train_gnns.py:332-333`).  All
parameters are re-homed as views into ONE contiguous fp32 buffer (and so are
their gradients), which is also what the DDP layer all-reduces in buckets.
"""
from __future__ import annotations

import math

import torch

from .lib import call


class FlatParams:
    """Re-homes ``params`` into one flat fp32 buffer; ``.grad`` of each parameter is
    a view into one flat gradient buffer (kept across steps, zeroed in one memset)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("isic_hip optimizers run on the MI355X only (no CPU fallback)")
        # 64-element (256-byte) alignment per tensor keeps every view 16-byte aligned
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64
        self.numel = n
        self.data = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            # keep each tensor's own dense layout (conv weights are channels_last = [O][Kh][Kw][I])
            v = torch.as_strided(self.data, p.shape, p.stride(), storage_offset=o)
            v.copy_(p.data)
            p.data = v
            p.grad = torch.as_strided(self.grad, p.shape, p.stride(), storage_offset=o)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):  # re-attach views if something replaced them
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = torch.as_strided(self.grad, p.shape, p.stride(), storage_offset=o)


class _FlatAdamBase:
    decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=None, flat=None):
        if weight_decay is None:
            weight_decay = 1e-2 if self.decoupled else 0.0
        self.flat = flat if flat is not None else FlatParams(list(params))
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        self.t = 0
        self.device_clock = None       # graphs.StepClock.tensor: t = clock[1] + 1 is then read on the device (captured steps)
        self.param_groups = [{"params": self.flat.params, "lr": self.lr, "betas": self.betas, "eps": self.eps,
                              "weight_decay": self.weight_decay}]

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self, grad_scale=1.0):
        g = self.param_groups[0]
        lr, (b1, b2), wd = float(g["lr"]), g["betas"], float(g["weight_decay"])
        if self.device_clock is not None:
            call("isic_adam_step_clk", self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.flat.numel, lr, b1,
                 b2, float(g["eps"]), (1.0 - lr * wd) if self.decoupled else 1.0, 0.0 if self.decoupled else wd,
                 float(grad_scale), None, self.device_clock)
            return
        self.t += 1
        bc1 = 1.0 - b1 ** self.t
        bc2_sqrt = math.sqrt(1.0 - b2 ** self.t)
        call("isic_adam_step", self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.flat.numel,
             lr / bc1, b1, b2, float(g["eps"]), (1.0 - lr * wd) if self.decoupled else 1.0,
             0.0 if self.decoupled else wd, bc2_sqrt, float(grad_scale), None)

    def sync_clock(self):
        """While a device step clock is attached the step count lives in HBM (clock[1]); bring the host copy up to date
        (called by ``state_dict`` and by ``graphs.StepClock.detach``: a checkpoint or a switch back to eager steps must
        continue Adam's bias correction from the steps actually taken)."""
        if self.device_clock is not None:
            self.t = int(self.device_clock[1].item())
        return self.t

    def state_dict(self):
        self.sync_clock()
        return {"t": self.t, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.t = int(sd["t"])
        if self.device_clock is not None:
            self.device_clock[1] = self.t
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0].update(sd["param_groups"][0])


class AdamW(_FlatAdamBase):
    decoupled = True


class Adam(_FlatAdamBase):
    decoupled = False
