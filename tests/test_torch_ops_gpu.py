"""``torch.ops.isic_hip.*`` (isic_hip/torch_ops.py): the registered custom ops against the ``autograd.Function`` surface the
modules use (same C-ABI launches -> bit-equal outputs and gradients), ``torch.library.opcheck`` (schema, fake tensors,
autograd registration, AOT dispatch) and a ``make_fx`` trace of a GraphMIL-shaped forward + backward in which the ops appear as
single nodes.  The Function surface itself is checked against the oracle in test_mil_gpu.py / test_graph_gpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from isic_hip import torch_ops  # noqa: F401  (registers the ops)
    return torch.ops.isic_hip


def _leaf(*shape, gen, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(DEV).requires_grad_(True)


def _same(a, b, what):
    assert a.shape == b.shape, what
    assert torch.equal(a, b), f"{what}: max |d| = {(a - b).abs().max().item():.3e}"


@pytest.mark.parametrize("act,p", [(0, 0.0), (1, 0.0), (2, 0.0), (1, 0.25)])
def test_linear_op_equals_function(act, p):
    from isic_hip import ops, torch_ops
    O = _ops()
    gen = torch.Generator().manual_seed(3)
    x, w, b = _leaf(300, 96, gen=gen), _leaf(72, 96, gen=gen, scale=0.1), _leaf(72, gen=gen)
    dy = torch.randn(300, 72, generator=gen).to(DEV)
    drop = ops.DropoutSpec(p, seed=7, stream=5) if p else None
    y0 = ops.linear(x, w, b, act=act, drop=drop)
    g0 = torch.autograd.grad(y0, (x, w, b), dy)
    y1 = O.linear(x, w, b, act, *torch_ops.drop_args(drop))
    g1 = torch.autograd.grad(y1, (x, w, b), dy)
    _same(y1, y0, "y")
    for a, r, n in zip(g1, g0, "xwb"):
        _same(a, r, "d" + n)
    # no bias, x not requiring grad
    y2 = O.linear(x.detach(), w, None, act, *torch_ops.drop_args(drop))
    (gw,) = torch.autograd.grad(y2, (w,), dy)
    (gw0,) = torch.autograd.grad(ops.linear(x.detach(), w, None, act=act, drop=drop), (w,), dy)
    _same(gw, gw0, "dw without bias")


def test_layer_norm_op_equals_function():
    from isic_hip import ops, torch_ops
    O = _ops()
    gen = torch.Generator().manual_seed(4)
    x, g, b, res = _leaf(500, 128, gen=gen), _leaf(128, gen=gen), _leaf(128, gen=gen), _leaf(500, 128, gen=gen)
    dy = torch.randn(500, 128, generator=gen).to(DEV)
    drop = ops.DropoutSpec(0.2, seed=9, stream=3)
    y0 = ops.layer_norm(x, g, b, eps=1e-5, relu=True, drop=drop, residual=res)
    g0 = torch.autograd.grad(y0, (x, g, b, res), dy)
    y1, mean, rstd = O.layer_norm(x, g, b, res, 1e-5, True, *torch_ops.drop_args(drop))
    g1 = torch.autograd.grad(y1, (x, g, b, res), dy)
    _same(y1, y0, "y")
    for a, r, n in zip(g1, g0, ("x", "gamma", "beta", "residual")):
        _same(a, r, "d" + n)
    assert mean.shape == (500,) and rstd.shape == (500,)


def _rand_graph_batch(gen, sizes, deg=6):
    from isic_hip.graph import GraphBatch
    offs = np.concatenate([[0], np.cumsum(sizes)])
    eis = [torch.stack([torch.randint(0, m, (deg * m,), generator=gen), torch.randint(0, m, (deg * m,), generator=gen)]) + int(o)
           for m, o in zip(sizes, offs[:-1])]
    return GraphBatch(torch.cat(eis, dim=1).to(DEV), int(offs[-1])), offs


def test_spmm_op_equals_function():
    from isic_hip import torch_ops
    from isic_hip.graph import spmm
    O = _ops()
    gen = torch.Generator().manual_seed(5)
    gb, offs = _rand_graph_batch(gen, [196] * 6 + [50, 77])
    n = int(offs[-1])
    x, bias = _leaf(n, 128, gen=gen), _leaf(128, gen=gen)
    dy = torch.randn(n, 128, generator=gen).to(DEV)
    y0 = spmm(x, gb, bias=bias)
    g0 = torch.autograd.grad(y0, (x, bias), dy)
    y1 = O.spmm(*torch_ops.graph_tensors(gb), x, bias, 1.0)
    g1 = torch.autograd.grad(y1, (x, bias), dy)
    _same(y1, y0, "A^x + b")
    _same(g1[0], g0[0], "dx")
    _same(g1[1], g0[1], "dbias")


@pytest.mark.parametrize("heads", [1, 4])
def test_attn_pool_op_equals_function(heads):
    from isic_hip import ops
    from isic_hip.bags import BagOffsets
    O = _ops()
    gen = torch.Generator().manual_seed(6)
    sizes = [196, 64, 1, 130, 17]
    offs = BagOffsets.from_lengths(sizes, DEV)
    T, H, A = offs.total, 128, 32
    h = _leaf(T, H, gen=gen)
    W2, b2 = _leaf(heads * A, H, gen=gen, scale=0.1), _leaf(heads * A, gen=gen, scale=0.1)
    w3, b3 = _leaf(heads, A, gen=gen, scale=0.3), _leaf(heads, gen=gen, scale=0.1)
    dz = torch.randn(len(sizes), H, generator=gen).to(DEV)
    z0, att0 = ops.attn_pool(h, W2, b2, w3, b3, offs.device, offs.max_bag, heads=heads)
    g0 = torch.autograd.grad(z0, (h, W2, b2, w3, b3), dz)
    z1, att1, _t = O.attn_pool(h, W2, b2, w3, b3, offs.device, offs.max_bag, heads)
    g1 = torch.autograd.grad(z1, (h, W2, b2, w3, b3), dz)
    _same(z1, z0, "z")
    _same(att1, att0, "att")
    for a, r, n in zip(g1, g0, ("h", "W2", "b2", "w3", "b3")):
        _same(a, r.reshape(a.shape), "d" + n)


def test_softmax_and_cross_entropy_ops_equal_functions():
    from isic_hip import ops
    O = _ops()
    gen = torch.Generator().manual_seed(7)
    x = _leaf(64, 7, gen=gen)
    lab = torch.randint(0, 7, (64,), generator=gen).to(DEV)
    dp = torch.randn(64, 7, generator=gen).to(DEV)
    p0 = ops.softmax_rows(x)
    p1 = O.softmax_rows(x)
    _same(p1, p0, "softmax")
    _same(torch.autograd.grad(p1, x, dp)[0], torch.autograd.grad(p0, x, dp)[0], "d softmax")
    for mode, inp in ((0, x), (1, p0.detach().requires_grad_(True))):
        l0 = ops.CrossEntropyFn.apply(inp, lab, mode)[0]
        l1, _d = O.cross_entropy(inp, lab, mode)
        _same(l1, l0, f"loss mode {mode}")
        _same(torch.autograd.grad(l1 * 3.0, inp)[0], torch.autograd.grad(l0 * 3.0, inp)[0], f"d loss mode {mode}")


def test_opcheck_schema_fake_autograd_and_aot_dispatch():
    """torch.library.opcheck on every differentiable op: the schema tells the truth about aliasing / mutation, the fake
    implementation matches the real one's metadata, autograd is registered through the dispatcher, and AOT dispatch (what
    torch.compile does first) reproduces eager."""
    from torch.library import opcheck
    from isic_hip.bags import BagOffsets
    from isic_hip import torch_ops
    O = _ops()
    gen = torch.Generator().manual_seed(8)
    tests = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_dynamic")
    x, w, b = _leaf(64, 32, gen=gen), _leaf(16, 32, gen=gen), _leaf(16, gen=gen)
    opcheck(O.linear.default, (x, w, b, 1, 0, 1.0, 0, 0), test_utils=tests)
    g, be = _leaf(32, gen=gen), _leaf(32, gen=gen)
    opcheck(O.layer_norm.default, (x, g, be, None, 1e-5, True, 0, 1.0, 0, 0), test_utils=tests)
    gb, offs = _rand_graph_batch(gen, [40, 60])
    xs, bs = _leaf(100, 32, gen=gen), _leaf(32, gen=gen)
    opcheck(O.spmm.default, (*torch_ops.graph_tensors(gb), xs, bs, 1.0), test_utils=tests)
    bo = BagOffsets.from_lengths([40, 60], DEV)
    W2, b2, w3, b3 = _leaf(16, 32, gen=gen, scale=0.1), _leaf(16, gen=gen), _leaf(2, 8, gen=gen), _leaf(2, gen=gen)
    opcheck(O.attn_pool.default, (xs, W2, b2, w3, b3, bo.device, bo.max_bag, 2), test_utils=tests)
    opcheck(O.softmax_rows.default, (x,), test_utils=tests)
    lab = torch.randint(0, 32, (64,), generator=gen).to(DEV)
    opcheck(O.cross_entropy.default, (x, lab, 0), test_utils=tests)


def test_graphmil_shaped_step_traces_to_registered_ops():
    """A GCN layer + LayerNorm join + attention pool + classifier + loss, forward AND backward, traced by make_fx: every
    launch group is one ``isic_hip.*`` node (no Python Function bodies in the graph), and running the traced graph gives the
    eager numbers."""
    from torch.fx.experimental.proxy_tensor import make_fx
    from isic_hip.bags import BagOffsets
    from isic_hip import torch_ops
    O = _ops()
    gen = torch.Generator().manual_seed(9)
    sizes = [196] * 4
    gb, offs = _rand_graph_batch(gen, sizes, deg=8)
    bo = BagOffsets.from_lengths(sizes, DEV)
    csr = torch_ops.graph_tensors(gb)
    n, F, C = bo.total, 128, 4
    P = [_leaf(F, F, gen=gen, scale=0.08), _leaf(F, gen=gen, scale=0.1), _leaf(F, gen=gen), _leaf(F, gen=gen),
         _leaf(64, F, gen=gen, scale=0.1), _leaf(64, gen=gen, scale=0.1), _leaf(2, 32, gen=gen, scale=0.3),
         _leaf(2, gen=gen, scale=0.1), _leaf(C, F, gen=gen, scale=0.1), _leaf(C, gen=gen, scale=0.1)]
    x = torch.randn(n, F, generator=gen).to(DEV)
    lab = torch.randint(0, C, (len(sizes),), generator=gen).to(DEV)

    def step(x, lab, *p):
        W, bconv, g, be, W2, b2, w3, b3, Wc, bc = p
        h = O.linear(x, W, None, 0, 0, 1.0, 0, 0)
        h = O.spmm(*csr, h, bconv, 1.0)
        h, _m, _r = O.layer_norm(h, g, be, x, 1e-5, True, 1 << 30, 4.0 / 3.0, 11, 2)
        z, _att, _t = O.attn_pool(h, W2, b2, w3, b3, bo.device, bo.max_bag, 2)
        prob = O.softmax_rows(O.linear(z, Wc, bc, 0, 0, 1.0, 0, 0))
        loss, _d = O.cross_entropy(prob, lab, 1)
        return (loss,) + torch.autograd.grad(loss, p)

    eager = step(x, lab, *P)
    gm = make_fx(step)(x, lab, *P)
    targets = [str(nd.target) for nd in gm.graph.nodes if nd.op == "call_function"]
    ours = [t for t in targets if t.startswith("isic_hip.")]
    for name in ("linear", "spmm", "layer_norm", "attn_pool", "softmax_rows", "cross_entropy", "linear_backward",
                 "spmm_backward", "layer_norm_backward", "attn_pool_backward", "softmax_rows_backward"):
        assert any(t.split(".")[1] == name for t in ours), (name, ours)
    traced = gm(x, lab, *P)
    for i, (a, r) in enumerate(zip(traced, eager)):
        _same(a, r, f"traced output {i}")
    assert torch.isfinite(eager[0]) and all(torch.isfinite(t).all() for t in eager[1:])
