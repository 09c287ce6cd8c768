"""World-size-2 gloo test (CPU) of the bucketed gradient exchange used for the DDP
train step: the bucketed, overlapped all-reduce over the flat gradient buffer must
equal a plain sum over ranks, and sharding must cover every bag exactly once."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-isic_amd"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isic_hip.ddp import GradSync, shard_range
    n = 10_000
    g = torch.Generator().manual_seed(rank)
    buf = torch.randn(n, generator=g)
    expect = sum(torch.randn(n, generator=torch.Generator().manual_seed(r)) for r in range(world))
    sync = GradSync(buf, world_size=world, bucket_bytes=4 * 3000)
    # backward order: tail of the buffer first
    for lo in (9000, 7500, 6900, 3000, 2500, 10):
        sync.mark_ready(lo)
    launched = sync.finish()
    ok = torch.allclose(buf, expect, atol=1e-5)
    covered = sorted(launched)
    contiguous = covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    big_enough = all(hi - lo >= 3000 for lo, hi in launched[:-1])
    lo, hi = shard_range(11, rank, world)
    q.put((rank, ok, contiguous, big_enough, len(launched), (lo, hi)))
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    for rank, ok, contiguous, big_enough, n_launch, shard in res:
        assert ok and contiguous and big_enough, (rank, ok, contiguous, big_enough)
        assert n_launch == 3          # [7500,10000) [3000,7500) [0,3000)
    assert res[0][5] == (0, 6) and res[1][5] == (6, 11)


def _encoder_mark_offsets():
    """Flat-buffer offsets of the encoder's `grad_ready_hook` calls, in the order a real backward makes them
    (tests/test_ddp_gpu.py records the same sequence on the MI355X): per residual block, last block first, the
    offset of the block's FIRST parameter; then the stem (offset 0).  Same 64-element alignment as FlatParams."""
    from isic_hip.encoder import ResNet18Encoder
    enc = ResNet18Encoder()
    offs, n = {}, 0
    for name, p in enc.named_parameters():
        offs[name] = n
        n += (p.numel() + 63) // 64 * 64
    marks = []
    for pre, ds in reversed(enc.blocks):
        names = [k for k in offs if k.startswith(pre + ".")]
        marks.append(min(offs[k] for k in names))
    marks.append(0)
    return marks, n


def _worker_encoder(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isic_hip.ddp import GradSync
    marks, n = _encoder_mark_offsets()
    g = torch.Generator().manual_seed(100 + rank)
    buf = torch.randn(n, generator=g)
    expect = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    sync = GradSync(buf, world_size=world)                          # the shipped 16 MiB buckets
    for lo in marks:
        sync.mark_ready(lo)
    launched = sync.finish()
    cover = sorted(launched)
    ok = torch.allclose(buf, expect, atol=1e-5)
    tiled = cover[0][0] == 0 and cover[-1][1] == n and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    q.put((rank, ok, tiled, launched, marks == sorted(marks, reverse=True)))
    dist.destroy_process_group()


def test_bucketed_allreduce_in_encoder_backward_order():
    """Two gloo ranks, the flat buffer laid out as ResNet-18's parameters and `mark_ready` driven in the order
    the encoder's backward fires it: every element is summed exactly once, in 3-4 collectives of >= 16 MiB."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_encoder, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, tiled, launched, desc in res:
        assert ok and tiled and desc, (rank, ok, tiled, desc)
        assert 2 <= len(launched) <= 4 and all(hi - lo >= (16 << 20) // 4 for lo, hi in launched[:-1])
    assert res[0][3] == res[1][3]


def test_gradsync_single_process_is_noop():
    from isic_hip.ddp import GradSync
    buf = torch.arange(100, dtype=torch.float32)
    s = GradSync(buf, world_size=1, bucket_bytes=4 * 30)
    for lo in (80, 60, 20):
        s.mark_ready(lo)
    launched = s.finish()
    assert sorted(launched) == [(0, 20), (20, 60), (60, 100)]
    assert torch.equal(buf, torch.arange(100, dtype=torch.float32))
