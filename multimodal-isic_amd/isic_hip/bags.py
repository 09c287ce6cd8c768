"""CSR description of a batch of ragged bags / graphs: ``offsets[B+1]`` on the
device (int64, what the kernels read) plus the host-side facts the launch needs
(number of bags, longest bag) so that no device->host sync is ever required."""
from __future__ import annotations

import numpy as np
import torch

_SINGLE_CACHE = {}
_UNIFORM_CACHE = {}


class BagOffsets:
    __slots__ = ("device", "host", "num_bags", "max_bag", "total")

    def __init__(self, host_offsets, device):
        host = np.asarray(host_offsets, dtype=np.int64).reshape(-1)
        if host.size < 1 or host[0] != 0 or np.any(np.diff(host) < 0):
            raise ValueError("offsets must start at 0 and be non-decreasing")
        self.host = host
        self.num_bags = int(host.size - 1)
        self.max_bag = int(np.diff(host).max()) if host.size > 1 else 0
        self.total = int(host[-1])
        self.device = torch.from_numpy(host).to(device)

    @classmethod
    def single(cls, n, device):
        key = (int(n), str(device))
        o = _SINGLE_CACHE.get(key)
        if o is None:
            o = _SINGLE_CACHE[key] = cls([0, int(n)], device)
        return o

    @classmethod
    def uniform(cls, num_bags, bag_size, device):
        """Equal-sized bags; cached per (count, size, device): a training loop asks for the same offsets every step
        and each construction is a host -> device copy."""
        key = (int(num_bags), int(bag_size), str(device))
        o = _UNIFORM_CACHE.get(key)
        if o is None:
            if len(_UNIFORM_CACHE) > 64:
                _UNIFORM_CACHE.clear()
            o = _UNIFORM_CACHE[key] = cls(np.arange(num_bags + 1, dtype=np.int64) * int(bag_size), device)
        return o

    @classmethod
    def from_lengths(cls, lengths, device):
        return cls(np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))]), device)


def as_offsets(offsets, device):
    if isinstance(offsets, BagOffsets):
        return offsets
    if isinstance(offsets, torch.Tensor):
        return BagOffsets(offsets.detach().cpu().numpy(), device)  # host copy: prefer passing BagOffsets
    return BagOffsets(offsets, device)
