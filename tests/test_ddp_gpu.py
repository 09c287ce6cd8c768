"""Data-parallel evidence on ONE MI355X (an 8-GPU node is not available to the build): the gradient
exchange is driven from a REAL backward of the flagship model, and sharding a step's bags over ranks
is shown to give the single-process gradients ("sum of shard grads == single-rank batched
grads").  The two-rank collective itself is covered on CPU (tests/test_ddp_cpu.py, gloo)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_gradsync_buckets_follow_a_real_backward():
    """`GradSync.mark_ready(lo)` promises "every gradient at flat offset >= lo is final".  With the
    encoder's `grad_ready_hook` installed on a real MultiModalMILNet: (i) the hook fires with
    non-increasing offsets and the launched slices tile [0, numel) exactly once, tail first; (ii) at every
    `mark_ready(lo)` a snapshot of grad[lo:] (taken on the stream the hook runs on) equals the FINAL
    grad[lo:] after backward -- in particular autograd has finished every head gradient before the
    encoder's backward starts, which the overlap relies on."""
    from isic_hip import ddp, optim
    from model import MultiModalMILNet
    torch.manual_seed(0)
    net = MultiModalMILNet(hidden_dim=32, att_dim=16, dropout=0.0, radiomics_dim=16, num_classes=7).to(DEV)
    net.train()
    opt = optim.AdamW(net.parameters(), lr=1e-3)
    flat = opt.flat
    sync = ddp.GradSync(flat.grad, world_size=1, bucket_bytes=4 << 20)      # 11.2 M parameters -> ~11 buckets
    ddp.attach(net.encoder, flat, sync)
    marks, snaps = [], []
    orig = sync.mark_ready

    def spy(lo):
        lo = max(0, min(int(lo), flat.numel))
        marks.append(lo)
        snaps.append((lo, flat.grad[lo:].clone()))
        orig(lo)
    sync.mark_ready = spy
    B, K, S = 4, 3, 64
    g = torch.Generator(device=DEV).manual_seed(1)
    img = torch.randn(B, K, 3, S, S, device=DEV, generator=g)
    rad = torch.randn(B, 16, device=DEV, generator=g)
    y = torch.arange(B, device=DEV) % 7
    opt.zero_grad()
    sync.reset()
    net.loss(net(img, rad), y).backward()
    launched = sync.finish()
    torch.cuda.synchronize()
    assert len(marks) == 9 and marks == sorted(marks, reverse=True) and marks[-1] == 0      # 8 residual blocks + stem
    assert launched == sorted(launched, reverse=True)                                       # tail of the buffer first
    cover = sorted(launched)
    assert cover[0][0] == 0 and cover[-1][1] == flat.numel and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert all(hi - lo >= (4 << 20) // 4 for lo, hi in launched[:-1]) and len(launched) >= 3
    for lo, snap in snaps:
        assert torch.equal(snap, flat.grad[lo:]), f"gradients at offsets >= {lo} changed after they were marked final"
    assert float(flat.grad.abs().sum()) > 0 and bool(torch.isfinite(flat.grad).all())


@pytest.mark.parametrize("kind", ["teacher", "graphmil"])
def test_shard_gradients_sum_to_batched_gradients(kind):
    """No BatchNorm in the MIL head / GraphMIL: a step over B bags == the sum of the steps over its two rank
    shards with the loss scaled as the train loops do (local mean * local / global count, train.py), to 1e-5
    of the gradient scale.  (The encoder's BatchNorm uses per-rank statistics: DESIGN.md 5.)"""
    import build_graphs as bg
    from dataset import synthetic_latent_bags
    from gnn_models import GraphMIL
    from isic_hip import ddp, ops, train as T
    from isic_hip.bags import BagOffsets
    from utils_g_mil import AttentionMIL_teacher
    torch.manual_seed(2)
    n, N, D = 10, 24, 32
    bags, labels = synthetic_latent_bags(n, N, D, classes=7, shift=0.7, seed=4)
    if kind == "teacher":
        model = AttentionMIL_teacher(D, 16, 8, dropout=0.0, num_classes=7).to(DEV)
        store = T.BagStore(bags, torch.device(DEV))

        def loss_of(idx, scale):
            x, offs = store.batch(idx)
            y = torch.as_tensor(labels[idx], device=DEV)
            return ops.cross_entropy(model(x, offs)["bag_logits"], y) * scale
    else:
        model = GraphMIL(D, "gcn", 16, 2, 0.0, att_dim=8, att_heads=4, pool_dropout=0.0, classifier_dim=12,
                         classifier_light=True, num_classes=7).to(DEV)
        recs = [{"x": b, "edge_index": bg._knn_edge_index(torch.from_numpy(b), 3).numpy(), "y": int(y)}
                for b, y in zip(bags, labels)]
        store = T.GraphStore(recs, torch.device(DEV), True, mode=model.graph_mode)

        def loss_of(idx, scale):
            x, offs, g = store.batch(idx)
            probs, _ = model(x, offsets=offs, graph=g)
            return ops.cross_entropy_from_probs(probs, torch.as_tensor(store.y[idx], device=DEV)) * scale
    model.train()
    glob = [7, 1, 4, 9, 0, 3, 8]                                   # one step's bags, odd count: shards of 4 and 3

    def grads(parts):
        for p in model.parameters():
            p.grad = None
        for idx, scale in parts:
            loss_of(idx, scale).backward()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    full = grads([(glob, 1.0)])
    world = 2
    parts = []
    for r in range(world):
        lo, hi = ddp.shard_range(len(glob), r, world)
        mine = glob[lo:hi]
        parts.append((mine, len(mine) * world / len(glob)))         # the train loops' scaling; all-reduce SUM, then / world
    shard = grads(parts)
    for k in full:
        a, b = shard[k] / world, full[k]
        tol = 1e-5 * float(b.abs().max()) + 1e-8
        assert float((a - b).abs().max()) <= tol, (k, float((a - b).abs().max()), tol)


def test_two_rank_rehearsal_of_the_bench_step_on_one_gpu(tmp_path):
    """The whole multi-rank bench path -- rendezvous, per-rank sharding, the bucketed gradient exchange fired by the
    real encoder backward, identical AdamW on every rank, buffer averaging, the MAX-over-ranks timing -- with two
    processes that share cuda:0 and gloo collectives (`bench.py --rehearse-on-one-gpu`).  RCCL itself needs two GPUs."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "2",
           "--warmup", "1", "--bags-per-step", "2", "--patches", "4", "--image-size", "64", "--no-cpu-baseline", "--no-sublines"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_bags_per_step"] == 4
    assert line["value"] > 0 and line["config"]["final_loss"] == line["config"]["final_loss"]      # finite, not NaN
    assert "rehearsal" in line


def test_bare_multi_gpu_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun (VERDICT r2 missing #2): the parent, which never touches the GPU,
    starts one child per rank with the torchrun environment and relays rank 0's line -- here as the one-GPU rehearsal
    (gloo), with the configs[3] / configs[4] sub-lines attached at toy size."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "2", "--warmup", "1",
           "--bags-per-step", "2", "--patches", "4", "--image-size", "64", "--no-cpu-baseline", "--sub-steps", "2",
           "--graphs-per-step", "8", "--images-per-step", "8", "--teacher-bags-per-step", "8", "--knn-graphs-per-step", "16", "--pipeline-images", "48"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["value"] > 0
    assert line["gnn"]["n_gpus"] == 2 and line["gnn"]["value"] > 0 and line["gnn"]["roofline"]["bound"] == "hbm"
    assert line["vit"]["n_gpus"] == 2 and line["vit"]["value"] > 0 and line["vit"]["roofline"]["bound"] == "hbm"
    assert line["teacher"]["n_gpus"] == 2 and line["teacher"]["value"] > 0 and line["teacher"]["tuned"]["value"] > 0
    assert line["knn"]["n_gpus"] == 2 and line["knn"]["value"] > 0 and line["knn"]["roofline"]["bound"] == "mfma"
    assert "hipGraph" in line["gnn"]["config"]["step_launch"] and "hipGraph" in line["teacher"]["config"]["step_launch"]
    assert line["pipeline"]["n_gpus"] == 2 and line["pipeline"]["value"] > 0 and len(line["pipeline"]["seconds_per_stage"]) == 5


def test_captured_gnn_step_under_data_parallelism_equals_the_eager_step():
    """configs[3] under DDP: with world size > 1 the step is TWO hipGraph replays around one eager
    bucketed all-reduce ([forward + backward] -> GradSync.finish() -> [AdamW + clock.advance()]) instead of ~80 eagerly
    enqueued launches.  Two ranks on one GPU (gloo collectives): the captured run must not fall back to the eager step and
    must leave the same parameters as the eager 2-rank run of the same steps (same draws, same dropout words; the AdamW
    step size of the clock form differs by at most 1 ulp, graphs.py)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    base = [sys.executable, os.path.join(root, "bench.py"), "--config", "gnn", "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "4",
            "--warmup", "1", "--graphs-per-step", "16", "--no-cpu-baseline"]
    lines = []
    for extra in ([], ["--no-graph"]):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=900, cwd=root, env=env)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
        lines.append(json.loads(r.stdout.strip().splitlines()[-1]))
    cap, eager = lines
    assert cap["n_gpus"] == 2 and "two hipGraph replays" in cap["config"]["step_launch"]
    assert eager["config"]["step_launch"] == "eager"
    a, b = cap["config"]["param_l1_after_run"], eager["config"]["param_l1_after_run"]
    assert abs(a - b) <= 2e-6 * abs(b), (a, b)
    assert abs(cap["config"]["final_loss"] - eager["config"]["final_loss"]) <= 1e-5 * max(1.0, abs(eager["config"]["final_loss"]))


_RCCL_CHILD = r"""
import os, sys, json
import torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, "multimodal-isic_amd")); sys.path.insert(0, root)
from isic_hip import ddp, optim
from model import MultiModalMILNet
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # "nccl" IS RCCL on ROCm
assert dist.get_backend() == "nccl"
torch.manual_seed(0)
net = MultiModalMILNet(hidden_dim=32, att_dim=16, dropout=0.0, radiomics_dim=16, num_classes=7).to(dev)
net.train()
opt = optim.AdamW(net.parameters(), lr=1e-3)
flat = opt.flat
g = torch.Generator(device=dev).manual_seed(1)
B, K, S = 4, 3, 64
img = torch.randn(B, K, 3, S, S, device=dev, generator=g)
rad = torch.randn(B, 16, device=dev, generator=g)
y = torch.arange(B, device=dev) % 7
ddp.broadcast_parameters(flat.data)
out = {}

def backward(force, side):
    sync = ddp.GradSync(flat.grad, world_size=1, bucket_bytes=4 << 20, force_collectives=force)
    ddp.attach(net.encoder, flat, sync)
    net.encoder.wgrad_stream = side
    snaps = []
    orig = sync.mark_ready
    def spy(lo):
        lo = max(0, min(int(lo), flat.numel))
        snaps.append((lo, flat.grad[lo:].clone()))       # on the stream the hook runs on, BEFORE the collective
        orig(lo)
    sync.mark_ready = spy
    opt.zero_grad(); sync.reset()
    net.set_dropout_state(seed=5, step=0)               # the projection / fusion MLPs drop out in train mode: same words every time
    net.loss(net(img, rad), y).backward()
    launched = sync.finish()
    torch.cuda.synchronize()
    return flat.grad.clone(), snaps, launched, len(sync.work)

ref, _, _, _ = backward(False, False)
for side in (False, True):
    got, snaps, launched, _ = backward(True, side)
    key = "side" if side else "main"
    # world 1: the all-reduce is an identity, so what it leaves behind must be what backward wrote ...
    ok_snap = all(bool(torch.equal(s, got[lo:])) for lo, s in snaps)
    # ... and equal to a backward without any collective.  Only the ENCODER's gradients are compared bit for bit (the
    # encoder step is bit-reproducible since round 3; the few head gradients go through split-K fp32 atomics)
    enc_lo = min(o for p_, o in zip(flat.params, flat.offsets) if any(p_ is q for q in net.encoder.parameters()))
    enc_hi = max(o + p_.numel() for p_, o in zip(flat.params, flat.offsets) if any(p_ is q for q in net.encoder.parameters()))
    out.setdefault("encoder_equal", True)
    out["encoder_equal"] = out["encoder_equal"] and bool(torch.equal(got[enc_lo:enc_hi], ref[enc_lo:enc_hi]))
    rel = float((got - ref).norm() / ref.norm())
    cover = sorted(launched)
    tiled = cover[0][0] == 0 and cover[-1][1] == flat.numel and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    out[key] = {"snapshots_equal_final": ok_snap, "rel_vs_no_collective": rel, "collectives": len(launched), "tiled": tiled}
# the next step is unaffected: parameters update and a further backward gives finite, non-zero gradients
opt.step(grad_scale=1.0)
nxt, _, _, _ = backward(True, True)
out["next_step_finite"] = bool(torch.isfinite(nxt).all()) and float(nxt.abs().sum()) > 0
t = torch.ones(1 << 20, device=dev)
dist.all_reduce(t)
out["plain_allreduce_identity"] = bool((t == 1).all())
dist.barrier()
dist.destroy_process_group()
print("RCCL_RESULT " + json.dumps(out), flush=True)
"""


def test_rccl_collectives_from_a_real_backward_world_size_1(tmp_path):
    """RCCL on the hardware this build can reach (one GPU): a child process initialises the "nccl" (= RCCL) process group
    with world size 1 and drives ``GradSync`` with collectives FORCED on from the real MultiModalMILNet backward, with the
    weight gradients on the main stream and on the side stream.  World 1 makes every all-reduce an identity, so the
    reduced buffer must equal the gradient backward wrote (snapshot at ``mark_ready``, and a backward without
    collectives), which exercises what gloo rehearsals cannot: librccl loading, communicator creation, and the ordering
    between the backward stream, the weight-gradient stream and RCCL's own stream.  The only RCCL evidence until the
    driver's 8-GPU SCALE run."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rccl_child.py"
    script.write_text(_RCCL_CHILD)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RCCL_RESULT ")][-1][len("RCCL_RESULT "):])
    for key in ("main", "side"):
        assert res[key]["snapshots_equal_final"], res
        assert res[key]["tiled"] and res[key]["collectives"] >= 3, res
        assert res[key]["rel_vs_no_collective"] < 1e-4, res      # head gradients: fp32 atomics order only; a clobbered bucket reads ~1
    assert res["encoder_equal"], res
    assert res["next_step_finite"] and res["plain_allreduce_identity"], res
