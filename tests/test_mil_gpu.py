"""GPU parity of the attention-MIL heads (HIP path through the C ABI) against the
reference-generated golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import assert_close, check_grad, formula_params, load_golden
from oracle import formula, mil, philox

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = ["small", "ref", "tuned", "rag1", "rag5", "rag64", "rag196"]


def _teacher(g, dropout=0.5):
    import utils_g_mil
    N, D, H, A, C = (int(v) for v in g["dims"])
    m = utils_g_mil.AttentionMIL_teacher(D, H, A, dropout, C)
    m.load_state_dict(formula_params(g))
    return m.to(DEV), (N, D, H, A, C)


@pytest.mark.parametrize("tag", CASES)
def test_teacher_golden(tag):
    g = load_golden(f"teacher_{tag}.npz")
    m, (N, D, H, A, C) = _teacher(g)
    m.eval()
    x = formula.formula_input(N, D).to(DEV).requires_grad_(True)
    out = m(x)
    # fp32 tolerance: exact-fp32 MFMA, only the summation order differs from torch CPU
    for k in ("bag_logits", "bag_probs", "attention", "patch_logits", "patch_probs"):
        assert_close(out[k], g[f"out.{k}"], rtol=3e-5, atol=2e-6, what=k)
    from isic_hip import ops
    loss = ops.cross_entropy(out["bag_logits"].unsqueeze(0), torch.from_numpy(g["label"]).to(DEV))
    assert_close(loss, g["loss"], rtol=3e-5, what="loss")
    loss.backward()
    for k, p in m.named_parameters():
        check_grad(g, k, p.grad, rtol=3e-4, atol=2e-6)
    check_grad(g, "x", x.grad, rtol=3e-4, atol=2e-6)


@pytest.mark.parametrize("tag", CASES)
def test_attention_mil_golden(tag):
    import utils_g_mil
    g = load_golden(f"attmil_{tag}.npz")
    N, D, H, A, C = (int(v) for v in g["dims"])
    m = utils_g_mil.AttentionMIL(D, H, A, 0.5, C)
    m.load_state_dict(formula_params(g))
    m = m.to(DEV).eval()
    probs, a = m(formula.formula_input(N, D).to(DEV))
    assert_close(probs, g["probs"], rtol=3e-5, atol=2e-6, what="probs")
    assert_close(a, g["a"], rtol=3e-5, atol=2e-6, what="a")


def test_ragged_batch_matches_oracle_with_grads():
    g = load_golden("teacher_small.npz")
    m, (N, D, H, A, C) = _teacher(g)
    m.eval()
    lens = [3, 1, 7, 0, 5, 64, 196, 2]
    offs = np.concatenate([[0], np.cumsum(lens)])
    x = formula.formula_input(int(offs[-1]), D, phase=1.3)
    y = torch.tensor([i % C for i in range(len(lens))])
    p = formula_params(g)
    loss_o, out_o, grads_o = mil.teacher_loss_and_grads(p, x, y, offsets=offs)
    xd = x.to(DEV).requires_grad_(True)
    out = m(xd, offs)
    from isic_hip import ops
    loss = ops.cross_entropy(out["bag_logits"], y.to(DEV))
    loss.backward()
    for k in ("bag_logits", "bag_probs", "attention", "patch_logits", "patch_probs"):
        assert_close(out[k], out_o[k], rtol=3e-5, atol=2e-6, what=k)
    assert_close(loss, loss_o, rtol=3e-5)
    for k, prm in m.named_parameters():
        assert_close(prm.grad, grads_o[k], rtol=3e-4, atol=2e-6, what=k)
    assert_close(xd.grad, grads_o["x"], rtol=3e-4, atol=2e-6, what="x")


def test_dropout_matches_counter_based_oracle():
    g = load_golden("teacher_small.npz")
    m, (N, D, H, A, C) = _teacher(g, dropout=0.71)
    m.train()
    m.set_dropout_state(seed=1234, step=5)
    lens = [16, 9, 33]
    offs = np.concatenate([[0], np.cumsum(lens)])
    x = formula.formula_input(int(offs[-1]), D, phase=0.2)
    out = m(x.to(DEV), offs, return_pooled=True)
    drop = {"p": 0.71, "seed": 1234, "stream": 5 * 1024 + 0}
    out_o = mil.teacher_forward_batched(formula_params(g), x, offs, drop=drop)
    assert_close(out["hidden"], out_o["hidden"], rtol=3e-5, atol=2e-6, what="hidden")
    assert_close(out["bag_logits"], out_o["bag_logits"], rtol=3e-5, atol=2e-6, what="bag_logits")
    keep = philox.dropout_keep(out_o["hidden"].numel(), 0.71, 1234, 5 * 1024)
    assert 0.2 < keep.mean() < 0.4


def test_adamw_three_steps_golden():
    """01_train_mil_teacher.py:237-246 per-bag loop, HIP forward/backward + flat AdamW."""
    from isic_hip import ops, optim
    g = load_golden("teacher_adamw3.npz")
    m, (N, D, H, A, C) = _teacher(g, dropout=0.0)
    m.train()
    opt = optim.AdamW(m.parameters(), lr=float(g["lr"]), weight_decay=float(g["wd"]))
    for s in range(3):
        x = formula.formula_input(N, D, phase=0.5 + 0.3 * s).to(DEV)
        y = torch.tensor([s % C], device=DEV)
        opt.zero_grad()
        out = m(x)
        loss = ops.cross_entropy(out["bag_logits"].unsqueeze(0), y)
        loss.backward()
        opt.step()
        assert abs(float(loss.detach()) - float(g[f"loss{s}"])) < 2e-5
        for k, v in m.state_dict().items():
            if k == "attention.2.bias":
                # softmax is shift invariant: d loss / d b3 == 0 analytically, so Adam's first steps
                # (lr * g / (|g| + eps)) amplify pure rounding noise of either implementation.
                assert float(m.attention[2].bias.grad.abs().max()) < 1e-6
                continue
            assert_close(v, g[f"step{s}.{k}"], rtol=2e-5, atol=2e-7, what=f"step{s}.{k}")


def test_cpu_tensor_is_rejected():
    import utils_g_mil
    from isic_hip.lib import IsicHipError
    m = utils_g_mil.AttentionMIL_teacher(8, 4, 4, 0.0, 3)
    with pytest.raises(IsicHipError):
        m(torch.zeros(5, 8))


@pytest.mark.parametrize("shape", [(128, 200, 20000, True, False), (70, 768, 50176, True, False), (96, 64, 9000, False, True)])
@pytest.mark.parametrize("beta", [0.0, 1.0, 0.5])
def test_gemm_split_k_and_chunked_colsum(shape, beta):
    """Long reductions onto small outputs (the weight / bias gradients over all nodes of a graph batch) are split
    over K / over row chunks and accumulated with atomics onto the pre-scaled output: compared with fp64."""
    from isic_hip import ops
    M, N, K, ta, tb = shape
    g = torch.Generator().manual_seed(5)
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((N, K) if tb else (K, N), generator=g)
    c0 = torch.randn(M, N, generator=g)
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double() + beta * c0.double()
    out = c0.clone().to(DEV)
    ops.gemm(a.to(DEV), b.to(DEV), trans_a=ta, trans_b=tb, out=out, beta=beta)
    assert_close(out.cpu(), ref, rtol=2e-6, atol=1e-4, what=f"split-K gemm {shape} beta={beta}")
    x = torch.randn(K, N, generator=g)
    o0 = torch.randn(N, generator=g)
    o = o0.clone().to(DEV)
    ops.colsum(x.to(DEV), out=o, beta=beta)
    assert_close(o.cpu(), x.double().sum(0) + beta * o0.double(), rtol=2e-6, atol=1e-4, what=f"colsum {shape} beta={beta}")
