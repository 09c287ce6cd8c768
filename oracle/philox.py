"""Philox4x32-10 counter-based RNG in numpy (oracle / test infrastructure).

The reference draws dropout masks from torch's global generator
(`utils_g_mil.py:49-53`, `05_train_gnns.py:121,190`), whose CPU and GPU streams
differ, so a CPU-vs-GPU training comparison cannot share masks (SURVEY.md §7
"Dropout RNG").  The build therefore defines dropout with a counter-based
generator that is restated here bit-for-bit and implemented identically in
``csrc/common.h`` (`philox4x32_10`):

    element i of a dropout site uses word  (i & 3)  of
    Philox4x32-10(counter = (lo32(i>>2), hi32(i>>2), lo32(stream), hi32(stream)),
                  key     = (lo32(seed), hi32(seed)))
    keep(i)  <=>  word >= floor(p * 2**32)        (p in [0, 1))
    y = keep ? x * (1 / (1 - p)) : 0              (scale rounded to fp32 once)
"""
from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays ``c0..c3``; scalar keys."""
    c0 = c0.astype(np.uint64)
    c1 = c1.astype(np.uint64)
    c2 = c2.astype(np.uint64)
    c3 = c3.astype(np.uint64)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def random_u32(n, seed, stream):
    """First ``n`` 32-bit words of the (seed, stream) sequence."""
    nblk = (n + 3) // 4
    idx = np.arange(nblk, dtype=np.uint64)
    c0 = (idx & _MASK).astype(np.uint32)
    c1 = (idx >> np.uint64(32)).astype(np.uint32)
    c2 = np.full(nblk, int(stream) & 0xFFFFFFFF, dtype=np.uint32)
    c3 = np.full(nblk, (int(stream) >> 32) & 0xFFFFFFFF, dtype=np.uint32)
    r = philox4x32_10(c0, c1, c2, c3, int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    return np.stack(r, axis=1).reshape(-1)[:n]


def dropout_threshold(p):
    return int(np.floor(float(p) * 4294967296.0))


def dropout_keep(n, p, seed, stream):
    """Boolean keep-mask of ``n`` elements for drop probability ``p``."""
    if p <= 0.0:
        return np.ones(n, dtype=bool)
    return random_u32(n, seed, stream) >= np.uint32(min(dropout_threshold(p), 0xFFFFFFFF))


def dropout_scale(p):
    return np.float32(1.0) / (np.float32(1.0) - np.float32(p))
