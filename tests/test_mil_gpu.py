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


@pytest.mark.parametrize("dims", [(196, 96, 128, 64, 7), (50, 64, 136, 144, 5)])      # pool-kernel sums (H, A <= 128) / the wide path
def test_teacher_gradients_accumulated_in_place_equal_autograd_accumulation(dims):
    """`ops.fused_grad_accumulation`: the attention pool's backward adds dW2 / db2 / dw3 / db3 / dW4 / db4 straight into the
    flat gradient buffer (GEMM beta = 1, column sums with beta = 1 on column slices of the per-bag sums) instead of
    returning six tensors for six AccumulateGrad launches -- same gradients (1e-6 of their scale), also on top of a
    gradient that is already there."""
    import utils_g_mil
    from isic_hip import ops, optim
    from isic_hip.bags import BagOffsets
    K, D, H, A, C = dims
    torch.manual_seed(3)
    m = utils_g_mil.AttentionMIL_teacher(D, H, A, 0.3, C).to(DEV)
    m.train()
    opt = optim.AdamW(m.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(4)
    B = 9
    x = torch.randn(B * K, D, generator=gen).to(DEV)
    y = (torch.arange(B) % C).to(DEV)
    offs = BagOffsets.uniform(B, K, torch.device(DEV))

    def grads(fused, times):
        opt.zero_grad()
        for _ in range(times):
            m.set_dropout_state(seed=8, step=0)
            with ops.fused_grad_accumulation(fused):
                ops.cross_entropy(m(x, offs)["bag_logits"], y).backward()
        torch.cuda.synchronize()
        return opt.flat.grad.clone()
    for times in (1, 2):
        a, b = grads(False, times), grads(True, times)
        assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-9, (dims, times)
        assert float(a.abs().max()) > 0


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


# (M, N, K, trans_a, trans_b, bias, act): the graph path's products at a batch of 256 graphs (05_train_gnns.py:168-199) and
# edge cases of the persistent LDS-DMA kernel (gemm_f32p.hip): every operand layout, the swapped (C^T) roles, split-K,
# ragged M / N / K (multiples of 4 that are not multiples of the 256 x 128 x 32 tile), leading dimensions > width
PGEMM_CASES = [
    (50176, 128, 768, False, True, True, 1),       # input_proj forward: x W^T + b, ReLU
    (50176, 128, 128, False, True, True, 2),       # 128-wide layer forward, tanh (attention heads)
    (50176, 768, 128, False, False, False, 0),     # dX = dY W
    (128, 768, 50176, True, False, False, 0),      # dW = dY^T X: swapped roles, split-K over 50 176 nodes
    (512, 128, 50176, True, False, False, 0),      # dW of the concatenated attention heads: split-K, no swap
    (128, 128, 50176, True, False, False, 0),      # dW of a GCN layer
    (30000, 132, 100, False, True, True, 0),       # ragged everything, k contiguous x k contiguous
    (3004, 260, 2052, False, False, False, 0),     # ragged, k contiguous x k strided
    (1100, 520, 4100, True, True, True, 0),        # k strided x k contiguous with a bias (un-swapped)
    (260, 1028, 8200, True, False, False, 0),      # k strided x k strided, split-K, ragged
    # the row-panel kernel (gemm_f32r.hip: long M, K <= 128, weights resident in LDS) ...
    (30003, 192, 64, False, True, True, 1),        # ragged rows, a 128 + 64 column split, K = 64
    (20000, 64, 128, False, False, False, 0),      # one 64-column half, weights given [K][N]
    (4100, 128, 48, False, True, False, 2),        # K = 48
    # ... and the register-fed A^T B split-K kernel (gemm_f32t.hip: small output, long reduction)
    (64, 128, 50173, True, False, False, 0),       # K not a multiple of 4, a 64-row output
    (128, 256, 9001, True, False, False, 0),       # two column tiles
]


@pytest.mark.parametrize("case", PGEMM_CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}{'T' if c[3] else 'N'}{'T' if c[4] else 'N'}" for c in PGEMM_CASES])
@pytest.mark.parametrize("beta", [0.0, 1.0])
def test_persistent_fp32_gemm(case, beta):
    """`isic_gemm_f32_ws` on the shapes that take the persistent 256 x 128 x 32 kernel, the row-panel kernel or the
    register-fed A^T B kernel: against an fp64 product (exact-fp32
    MFMA: only the summation order differs, 2e-6 of the output scale per 1000 k), bit-identical between two runs (split-K
    partials are added in split order: no atomics), and with padded leading dimensions."""
    from isic_hip import ops
    from isic_hip.lib import call
    M, N, K, ta, tb, has_bias, act = case
    assert call("isic_gemm_f32_workspace_bytes", int(ta), int(tb), M, N, K) >= 0
    g = torch.Generator().manual_seed(M + N + K)
    pad_a, pad_b, pad_c = 8, 4, 12                  # leading dimensions wider than the matrices (strided row views)
    a_full = torch.randn((K, M + pad_a) if ta else (M, K + pad_a), generator=g).to(DEV)
    b_full = torch.randn((N, K + pad_b) if tb else (K, N + pad_b), generator=g).to(DEV)
    a = a_full[:, :M] if ta else a_full[:, :K]
    b = b_full[:, :K] if tb else b_full[:, :N]
    bias = torch.randn(N, generator=g).to(DEV) if has_bias else None
    c0 = torch.randn(M, N + pad_c, generator=g).to(DEV)

    def run():
        c = c0.clone()
        ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=bias, act=act, out=c[:, :N], beta=beta)
        torch.cuda.synchronize()
        return c
    c1, c2 = run(), run()
    assert torch.equal(c1, c2), "two runs of the same product differ: the split-K reduction must be order-fixed"
    assert torch.equal(c1[:, N:], c0[:, N:]), "wrote outside the output columns"
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
    if has_bias:
        ref = ref + bias.double()
    ref = torch.relu(ref) if act == 1 else torch.tanh(ref) if act == 2 else ref
    ref = ref + beta * c0[:, :N].double()
    tol = 2e-6 * max(1.0, K / 1000.0) * (1.0 if act != 2 else 4.0)
    assert_close(c1[:, :N], ref, rtol=tol, atol=tol * float(K) ** 0.5, what=f"persistent gemm {case} beta={beta}")


def test_multi_copy_gathers_and_accumulates_many_small_tensors_in_one_launch():
    """ops.multi_copy (isic_multi_copy_f32): up to 32 (dst, src) pairs per launch, copy or accumulate; 40 pairs exercise the chunking."""
    from isic_hip import ops
    g = torch.Generator().manual_seed(3)
    sizes = [1, 7, 128, 16384, 3, 4097] * 7
    srcs = [torch.randn(s, generator=g).to(DEV) for s in sizes[:40]]
    dsts = [torch.randn(s, generator=g).to(DEV) for s in sizes[:40]]
    before = [d.clone() for d in dsts]
    ops.multi_copy(dsts, srcs, accumulate=True)
    for d, b, s_ in zip(dsts, before, srcs):
        assert torch.equal(d, b + s_)
    ops.multi_copy(dsts, srcs)
    for d, s_ in zip(dsts, srcs):
        assert torch.equal(d, s_)
    with pytest.raises(Exception):
        ops.multi_copy(dsts[:2], srcs[:1])
