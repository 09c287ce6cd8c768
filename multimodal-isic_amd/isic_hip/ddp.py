"""Data-parallel gradient exchange for bags sharded over the GPUs of one node.

The reference trains single-process (`01_train_mil_teacher.py:235-246`); bags are
independent units, so the step shards by bag and needs exactly one exchange: a
sum all-reduce of the gradients (RCCL over xGMI; ``torch.distributed`` backend
"nccl" IS RCCL on ROCm; "gloo" on CPU for the tests).

All gradients live in ONE flat buffer (``optim.FlatParams``) in parameter
registration order, and backward produces them in reverse order (head first, then
the encoder from layer4 down to the stem).  ``GradSync.mark_ready(lo)`` says
"everything at flat offset >= lo is final"; whenever at least ``bucket_bytes`` of
final gradient has piled up, that contiguous slice is all-reduced asynchronously
(on the process group's own stream) while the remaining backward kernels keep
running.  xGMI is point-to-point (ring all-reduce is per-link bound, ~153 GB/s), so
buckets are large (default 16 MiB) and few: ResNet-18's 44.7 MB of fp32 gradient
goes out in 3-4 collectives.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, flat_grad, world_size=None, bucket_bytes=16 << 20, group=None, force_collectives=False):
        """``force_collectives``: launch the all-reduces even at world size 1 (an identity there) -- the way to put
        RCCL, its stream and the ordering against the backward / weight-gradient streams under test on ONE GPU."""
        self.buf = flat_grad
        self.group = group
        self.force = bool(force_collectives)
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.bucket_elems = max(1, int(bucket_bytes) // flat_grad.element_size())
        self.reset()

    def reset(self):
        self.hi = self.buf.numel()
        self.work = []
        self.launched = []   # (lo, hi) slices, for tests / logging

    def _launch(self, lo, hi):
        if hi <= lo:
            return
        self.launched.append((lo, hi))
        if self.world > 1 or self.force:
            self.work.append(dist.all_reduce(self.buf[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def mark_ready(self, lo):
        """Gradients at flat offsets >= ``lo`` are final."""
        lo = max(0, min(int(lo), self.hi))
        if self.hi - lo >= self.bucket_elems or (lo == 0 and self.hi > 0):
            self._launch(lo, self.hi)
            self.hi = lo

    def finish(self):
        """All-reduce whatever is left and wait for every collective."""
        if self.hi > 0:
            self._launch(0, self.hi)
            self.hi = 0
        for w in self.work:
            w.wait()
        out = self.launched
        self.reset()
        return out


def broadcast_parameters(flat_data, src=0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_data, src=src, group=group)


def average_buffers(module, group=None):
    """BatchNorm running statistics are updated from each rank's LOCAL batch (statistics are per rank, as in
    torch DDP without SyncBatchNorm), so they drift apart; before evaluation / checkpointing every float buffer
    is replaced by its mean over the ranks and integer buffers (``num_batches_tracked``) by rank 0's."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    world = dist.get_world_size(group)
    for b in module.buffers():
        if b.dtype.is_floating_point:
            dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
            b.div_(world)
        else:
            dist.broadcast(b, src=0, group=group)


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of ``n_items`` bags for ``rank`` (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def attach(model_encoder, flat, sync):
    """Wire an encoder's ``grad_ready_hook`` to ``sync``: a block is complete when its first
    parameter's gradient is; everything after it in the flat buffer was finished earlier."""
    offset_of = {id(p): o for p, o in zip(flat.params, flat.offsets)}
    name_to_off = {n: offset_of[id(p)] for n, p in model_encoder.named_parameters() if id(p) in offset_of}

    def hook(names):
        sync.mark_ready(min(name_to_off[n] for n in names if n in name_to_off))

    model_encoder.grad_ready_hook = hook
    return hook
