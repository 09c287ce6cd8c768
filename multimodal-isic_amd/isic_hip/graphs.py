"""Train steps captured into a hipGraph (``torch.cuda.CUDAGraph`` IS hipGraph on ROCm).

The reference runs one optimizer step per graph / bag from Python (`05_train_gnns.py:336-346`,
`01_train_mil_teacher.py:237-246`).  Batched on the MI355X, a GNN step is ~50 C-ABI launches plus ~30 small torch
kernels for ~1.3 ms of GPU work: the host needs longer to enqueue it than the GPU to run it.  Capturing the whole step
(forward, autograd backward, AdamW) removes Python from the loop; what a replay must still vary is handled on the device:

* the dropout stream id and Adam's step count come from a **device step clock** (``StepClock``: uint64[2] in HBM, read
  by the ``_clk`` C-ABI entries, advanced by the last kernel of the step) instead of host scalars frozen at capture;
* the step's inputs (graph indices, labels) live in static tensors the caller overwrites before ``replay()``.

With the clock at step s the captured step is the eager step s to the tolerance tests/test_train_gpu.py states (loss 1e-6,
parameters 2e-5: the clock form of AdamW forms lr / (1 - beta1^t) on the device in double precision from a double lr; the
eager form rounds the host's double quotient once -- the step sizes agree to 1 ulp of fp32).
"""
from __future__ import annotations

import torch

from .lib import call


class StepClock:
    """clock[0] = dropout step (stream id = base + clock[0] * 1024), clock[1] = optimizer steps taken."""

    def __init__(self, device, dropout_step=0, optimizer_steps=0):
        self.tensor = torch.tensor([int(dropout_step), int(optimizer_steps)], dtype=torch.int64, device=device)

    def advance(self, dropout_steps=1, optimizer_steps=1):
        call("isic_step_clock_advance", self.tensor, int(dropout_steps), int(optimizer_steps))

    def read(self):
        d, o = self.tensor.tolist()
        return int(d), int(o)

    def attach(self, model=None, optimizer=None):
        """Point every dropout clock of ``model`` (modules with a ``dropout_clock``) and ``optimizer`` at this clock."""
        if model is not None:
            for m in model.modules():
                clk = getattr(m, "dropout_clock", None)
                if clk is not None:
                    clk.device_clock = self.tensor
        if optimizer is not None:
            optimizer.device_clock = self.tensor
        return self

    @staticmethod
    def detach(model=None, optimizer=None):
        if model is not None:
            for m in model.modules():
                clk = getattr(m, "dropout_clock", None)
                if clk is not None:
                    clk.device_clock = None
        if optimizer is not None:
            if hasattr(optimizer, "sync_clock"):
                optimizer.sync_clock()               # the host step count continues from the steps the clock counted
            optimizer.device_clock = None


class CapturedStep:
    """``body()`` -- one whole train step reading only static tensors -- warmed up on a side stream and captured once;
    ``replay()`` runs it again.  ``body`` must end with ``clock.advance()`` when it uses a ``StepClock``.

    The warm-up runs ``body`` ``warmup`` times for real (allocator growth, one-time kernel attributes, autograd shapes):
    those are optimizer updates and clock advances on whatever the static inputs hold.  Pass ``optimizer`` and ``clock``
    (and ``buffers``: any other tensors the step mutates, e.g. BatchNorm running statistics) and the flat parameters, both
    Adam moments, the clock and the buffers are snapshotted before the warm-up and restored after it, so that the first
    ``replay()`` is step ``clock[1] + 1`` of the run; without them the warm-up CONSUMES ``warmup`` steps."""

    def __init__(self, body, warmup=3, optimizer=None, clock=None, buffers=()):
        self.body = body
        saved = []
        if optimizer is not None:
            saved += [optimizer.flat.data, optimizer.exp_avg, optimizer.exp_avg_sq]
        if clock is not None:
            saved.append(clock.tensor)
        saved += list(buffers)
        snap = [t.clone() for t in saved]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):          # allocator growth, one-time kernel attribute setup, autograd graph shapes
                body()
        torch.cuda.current_stream().wait_stream(side)
        for t, c in zip(saved, snap):
            t.copy_(c)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = body()

    def replay(self):
        self.graph.replay()
        return self.out
