"""The hot-path operators registered with ``torch.library``: ``torch.ops.isic_hip.*``.

Every op is a ``torch.library.custom_op`` over the same C-ABI launches the modules use (``isic_hip/ops.py``,
``isic_hip/graph.py`` -- ctypes into ``libisic_hip.so``), with a fake (meta) implementation for shape inference and an
autograd formula registered through ``register_autograd`` whose backward is itself a registered op -- so the ops are
opaque, traceable units for ``torch.compile`` / ``make_fx`` / ``torch.library.opcheck`` instead of Python
``autograd.Function`` bodies.  Arguments are plain tensors and scalars (a dropout site is ``(threshold, scale, seed,
stream)`` of ``ops.DropoutSpec``; a graph is its CSR tensors), as ``custom_op`` schemas require.

    y     = torch.ops.isic_hip.linear(x, W, b, act, thr, scale, seed, stream)          # nn.Linear (+ReLU/tanh, +dropout)
    y,m,r = torch.ops.isic_hip.layer_norm(x, gamma, beta, residual, eps, relu, thr, scale, seed, stream)
    out   = torch.ops.isic_hip.spmm(rowptr, col, val, rowptr_t, col_t, val_t, x, bias, alpha)  # GCNConv aggregation
    z,att = torch.ops.isic_hip.attn_pool(h, W2, b2, w3, b3, offsets, max_bag, heads)   # multi-head attention pool
    p     = torch.ops.isic_hip.softmax_rows(x)
    loss  = torch.ops.isic_hip.cross_entropy(inp, labels, mode)

Reference arithmetic: `utils_g_mil.py:49-97` (MIL head), `05_train_gnns.py:168-217` (GraphMIL forward), `model.py:74-83`.
The module classes keep calling the ``autograd.Function`` forms (same kernels, less dispatcher overhead per launch, and the
in-kernel gradient accumulation of ``ops.fused_grad_accumulation``); tests/test_torch_ops_gpu.py pins the two surfaces to
each other bit for bit.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.library import custom_op

from . import ops as _o
from .lib import call

_EMPTY = lambda t: t.new_empty((0,))        # placeholder for "gradient not needed" outputs (custom ops return tensors)


def _c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


# ----------------------------------------------------------------------------------------------- linear
@custom_op("isic_hip::linear", mutates_args=())
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], act: int, drop_threshold: int, drop_scale: float,
           drop_seed: int, drop_stream: int) -> Tensor:
    x2 = _c(x.reshape(-1, x.shape[-1]))
    w = _c(weight)
    b = _c(bias) if bias is not None else None
    if drop_threshold:
        if act != _o.ACT_RELU:
            raise ValueError("fused dropout is defined after ReLU only")
        y = _o.gemm(x2, w, trans_b=True, bias=b, act=_o.ACT_NONE)
        call("isic_relu_dropout_fwd_f32", y, y.numel(), int(drop_threshold), float(drop_scale), int(drop_seed), int(drop_stream))
    else:
        y = _o.gemm(x2, w, trans_b=True, bias=b, act=int(act))
    return y.reshape(*x.shape[:-1], w.shape[0])


@linear.register_fake
def _(x, weight, bias, act, drop_threshold, drop_scale, drop_seed, drop_stream):
    return x.new_empty((*x.shape[:-1], weight.shape[0]), dtype=torch.float32)


@custom_op("isic_hip::linear_backward", mutates_args=())
def linear_backward(dy: Tensor, x: Tensor, weight: Tensor, y: Tensor, act: int, drop_scale: float, need_dx: bool,
                    need_dw: bool, need_db: bool) -> Tuple[Tensor, Tensor, Tensor]:
    x2, w = _c(x.reshape(-1, x.shape[-1])), _c(weight)
    g = _c(dy.reshape(-1, w.shape[0]))
    if act == _o.ACT_RELU:
        g = g.clone()
        call("isic_relu_dropout_bwd_f32", _c(y.reshape(g.shape)), g, g.numel(), float(drop_scale))
    elif act == _o.ACT_TANH:
        g2 = torch.empty_like(g)
        call("isic_tanh_bwd_f32", g, _c(y.reshape(g.shape)), g2, g.numel())
        g = g2
    dx = _o.gemm(g, w).reshape(x.shape) if need_dx else _EMPTY(g)
    dw = _o.gemm(g, x2, trans_a=True) if need_dw else _EMPTY(g)
    db = _o.colsum(g) if need_db else _EMPTY(g)
    return dx, dw, db


@linear_backward.register_fake
def _(dy, x, weight, y, act, drop_scale, need_dx, need_dw, need_db):
    e = dy.new_empty((0,), dtype=torch.float32)
    return (x.new_empty(x.shape, dtype=torch.float32) if need_dx else e,
            weight.new_empty(weight.shape, dtype=torch.float32) if need_dw else e,
            weight.new_empty((weight.shape[0],), dtype=torch.float32) if need_db else e)


def _linear_setup(ctx, inputs, output):
    x, weight, bias, act, thr, scale, _seed, _stream = inputs
    ctx.act, ctx.scale, ctx.has_bias = int(act), (float(scale) if thr else 1.0), bias is not None
    ctx.save_for_backward(x, weight, output)


def _linear_bwd(ctx, dy):
    x, weight, y = ctx.saved_tensors
    need = ctx.needs_input_grad
    dx, dw, db = linear_backward(dy, x, weight, y, ctx.act, ctx.scale, need[0], need[1], ctx.has_bias and need[2])
    return (dx if need[0] else None, dw if need[1] else None, db if (ctx.has_bias and need[2]) else None, None, None, None,
            None, None)


linear.register_autograd(_linear_bwd, setup_context=_linear_setup)


# ----------------------------------------------------------------------------------------------- layer norm
@custom_op("isic_hip::layer_norm", mutates_args=())
def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, residual: Optional[Tensor], eps: float, relu: bool,
               drop_threshold: int, drop_scale: float, drop_seed: int, drop_stream: int) -> Tuple[Tensor, Tensor, Tensor]:
    x2 = _c(x.reshape(-1, x.shape[-1]))
    M, N = x2.shape
    res = _c(residual.reshape(-1, N)) if residual is not None else None
    y = torch.empty_like(x2)
    mean = torch.empty((M,), device=x2.device, dtype=torch.float32)
    rstd = torch.empty((M,), device=x2.device, dtype=torch.float32)
    call("isic_layernorm_fwd", x2, _c(gamma), _c(beta), res, y, mean, rstd, M, N, float(eps), int(relu), int(drop_threshold),
         float(drop_scale), int(drop_seed), int(drop_stream))
    return y.reshape(x.shape), mean, rstd


@layer_norm.register_fake
def _(x, gamma, beta, residual, eps, relu, drop_threshold, drop_scale, drop_seed, drop_stream):
    M = x.numel() // x.shape[-1]
    return (x.new_empty(x.shape, dtype=torch.float32), x.new_empty((M,), dtype=torch.float32),
            x.new_empty((M,), dtype=torch.float32))


@custom_op("isic_hip::layer_norm_backward", mutates_args=())
def layer_norm_backward(dy: Tensor, x: Tensor, gamma: Tensor, beta: Tensor, mean: Tensor, rstd: Tensor, relu: bool,
                        drop_threshold: int, drop_scale: float, drop_seed: int,
                        drop_stream: int) -> Tuple[Tensor, Tensor, Tensor]:
    x2 = _c(x.reshape(-1, x.shape[-1]))
    M, N = x2.shape
    dy2 = _c(dy.reshape(M, N))
    dx = torch.empty_like(x2)
    dg = torch.zeros((N,), device=x2.device, dtype=torch.float32)
    db = torch.zeros((N,), device=x2.device, dtype=torch.float32)
    ws = _o._workspace(call("isic_layernorm_bwd_workspace_bytes", N), x2.device)
    call("isic_layernorm_bwd_ws", dy2, x2, _c(gamma), _c(beta), mean, rstd, dx, dg, db, M, N, int(relu), int(drop_threshold),
         float(drop_scale), int(drop_seed), int(drop_stream), None, ws, ws.numel() if ws is not None else 0)
    return dx.reshape(x.shape), dg, db


@layer_norm_backward.register_fake
def _(dy, x, gamma, beta, mean, rstd, relu, drop_threshold, drop_scale, drop_seed, drop_stream):
    return (x.new_empty(x.shape, dtype=torch.float32), gamma.new_empty(gamma.shape, dtype=torch.float32),
            gamma.new_empty(gamma.shape, dtype=torch.float32))


def _ln_setup(ctx, inputs, output):
    x, gamma, beta, residual, _eps, relu, thr, scale, seed, stream = inputs
    _y, mean, rstd = output
    ctx.cfg = (bool(relu), int(thr), float(scale), int(seed), int(stream))
    ctx.has_res = residual is not None
    ctx.save_for_backward(x, gamma, beta, mean, rstd)


def _ln_bwd(ctx, dy, _dmean, _drstd):
    x, gamma, beta, mean, rstd = ctx.saved_tensors
    relu, thr, scale, seed, stream = ctx.cfg
    dx, dg, db = layer_norm_backward(dy, x, gamma, beta, mean, rstd, relu, thr, scale, seed, stream)
    return dx, dg, db, (dy if ctx.has_res else None), None, None, None, None, None, None


layer_norm.register_autograd(_ln_bwd, setup_context=_ln_setup)


# ----------------------------------------------------------------------------------------------- GCN aggregation
@custom_op("isic_hip::spmm", mutates_args=())
def spmm(rowptr: Tensor, col: Tensor, val: Tensor, rowptr_t: Tensor, col_t: Tensor, val_t: Tensor, x: Tensor,
         bias: Optional[Tensor], alpha: float) -> Tensor:
    x2 = _c(x)
    out = torch.empty_like(x2)
    call("isic_spmm_csr_f32", rowptr, col, val, x2, _c(bias) if bias is not None else None, out, x2.shape[0], x2.shape[1],
         float(alpha), None, 0.0)
    return out


@spmm.register_fake
def _(rowptr, col, val, rowptr_t, col_t, val_t, x, bias, alpha):
    return x.new_empty(x.shape, dtype=torch.float32)


@custom_op("isic_hip::spmm_backward", mutates_args=())
def spmm_backward(rowptr_t: Tensor, col_t: Tensor, val_t: Tensor, dy: Tensor, alpha: float, need_db: bool) -> Tuple[Tensor, Tensor]:
    g = _c(dy)
    dx = torch.empty_like(g)
    call("isic_spmm_csr_f32", rowptr_t, col_t, val_t, g, None, dx, g.shape[0], g.shape[1], float(alpha), None, 0.0)
    return dx, (_o.colsum(g) if need_db else _EMPTY(g))


@spmm_backward.register_fake
def _(rowptr_t, col_t, val_t, dy, alpha, need_db):
    return dy.new_empty(dy.shape, dtype=torch.float32), dy.new_empty((dy.shape[1],) if need_db else (0,), dtype=torch.float32)


def _spmm_setup(ctx, inputs, output):
    _rp, _c_, _v, rowptr_t, col_t, val_t, _x, bias, alpha = inputs
    ctx.alpha, ctx.has_bias = float(alpha), bias is not None
    ctx.save_for_backward(rowptr_t, col_t, val_t)


def _spmm_bwd(ctx, dy):
    rowptr_t, col_t, val_t = ctx.saved_tensors
    need_db = ctx.has_bias and ctx.needs_input_grad[7]
    dx, db = spmm_backward(rowptr_t, col_t, val_t, dy, ctx.alpha, need_db)
    return None, None, None, None, None, None, dx, (db if need_db else None), None


spmm.register_autograd(_spmm_bwd, setup_context=_spmm_setup)


# ----------------------------------------------------------------------------------------------- attention pool (GraphMIL form)
@custom_op("isic_hip::attn_pool", mutates_args=())
def attn_pool(h: Tensor, W2: Tensor, b2: Tensor, w3: Tensor, b3: Tensor, offsets: Tensor, max_bag: int,
              heads: int) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (z[B, H], att[T, heads], t[T, heads*A]); t = tanh(h W2^T + b2) is returned for the backward op."""
    h2 = _c(h)
    T, H = h2.shape
    A = W2.shape[0] // heads
    B = offsets.numel() - 1
    t = _o.gemm(h2, _c(W2), trans_b=True, bias=_c(b2), act=_o.ACT_TANH)
    att = torch.empty((T, heads), device=h2.device, dtype=torch.float32)
    z = torch.empty((B, H), device=h2.device, dtype=torch.float32)
    call("isic_attn_pool_fwd", h2, t, _c(w3).reshape(heads, A), _c(b3).reshape(heads), None, None, offsets, B, H, A, heads, 0,
         int(max_bag), att, z, None, None, None, None)
    return z, att, t


@attn_pool.register_fake
def _(h, W2, b2, w3, b3, offsets, max_bag, heads):
    T, H = h.shape
    return (h.new_empty((offsets.numel() - 1, H), dtype=torch.float32), h.new_empty((T, heads), dtype=torch.float32),
            h.new_empty((T, W2.shape[0]), dtype=torch.float32))


@custom_op("isic_hip::attn_pool_backward", mutates_args=())
def attn_pool_backward(dz: Tensor, h: Tensor, t: Tensor, att: Tensor, W2: Tensor, w3: Tensor, offsets: Tensor, max_bag: int,
                       heads: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    h2, W2c = _c(h), _c(W2)
    T, H = h2.shape
    A = W2c.shape[0] // heads
    B = offsets.numel() - 1
    d_h = torch.empty((T, H), device=h2.device, dtype=torch.float32)
    d_u = torch.empty((T, heads * A), device=h2.device, dtype=torch.float32)
    d_s = torch.empty((T, heads), device=h2.device, dtype=torch.float32)
    if H <= 128 and A <= 128:                                             # as ops.AttnPoolFn: per-bag sums out of the pool kernel
        nA = heads * A
        psum = torch.empty((B, 2 * nA + heads), device=h2.device, dtype=torch.float32)
        call("isic_attn_pool_bwd_sums", h2, t, att, None, _c(w3).reshape(heads, A), None, offsets, B, H, A, heads, 0, int(max_bag),
             None, _c(dz), d_h, 0, d_u, d_s, None, psum)
        sums = _o.colsum(psum)
        db2, dw3, db3 = sums[:nA].clone(), sums[nA:2 * nA].reshape(w3.shape).clone(), sums[2 * nA:].clone()
    else:
        call("isic_attn_pool_bwd", h2, t, att, None, _c(w3).reshape(heads, A), None, offsets, B, H, A, heads, 0, int(max_bag), None,
             _c(dz), d_h, 0, d_u, d_s, None)
        db2 = _o.colsum(d_u)
        full = _o.gemm(d_s, t, trans_a=True)                               # [heads, heads*A]
        dw3 = torch.stack([full[k, k * A:(k + 1) * A] for k in range(heads)]).reshape(w3.shape)
        db3 = _o.colsum(d_s)
    dW2 = _o.gemm(d_u, h2, trans_a=True)
    _o.gemm(d_u, W2c, out=d_h, beta=1.0)
    return d_h, dW2, db2, dw3, db3


@attn_pool_backward.register_fake
def _(dz, h, t, att, W2, w3, offsets, max_bag, heads):
    f = lambda s: h.new_empty(s, dtype=torch.float32)
    return f(h.shape), f(W2.shape), f((W2.shape[0],)), f(w3.shape), f((heads,))


def _pool_setup(ctx, inputs, output):
    h, W2, _b2, w3, _b3, offsets, max_bag, heads = inputs
    _z, att, t = output
    ctx.cfg = (int(max_bag), int(heads))
    ctx.b3_shape = inputs[4].shape
    ctx.save_for_backward(h, t, att, W2, w3, offsets)
    ctx.mark_non_differentiable(att, t)


def _pool_bwd(ctx, dz, _datt, _dt):
    h, t, att, W2, w3, offsets = ctx.saved_tensors
    max_bag, heads = ctx.cfg
    d_h, dW2, db2, dw3, db3 = attn_pool_backward(dz, h, t, att, W2, w3, offsets, max_bag, heads)
    return d_h, dW2, db2, dw3, db3.reshape(ctx.b3_shape), None, None, None


attn_pool.register_autograd(_pool_bwd, setup_context=_pool_setup)


# ----------------------------------------------------------------------------------------------- softmax / cross entropy
@custom_op("isic_hip::softmax_rows", mutates_args=())
def softmax_rows(x: Tensor) -> Tensor:
    x2 = _c(x.reshape(-1, x.shape[-1]))
    p = torch.empty_like(x2)
    call("isic_softmax_rows_fwd", x2, p, x2.shape[0], x2.shape[1])
    return p.reshape(x.shape)


@softmax_rows.register_fake
def _(x):
    return x.new_empty(x.shape, dtype=torch.float32)


@custom_op("isic_hip::softmax_rows_backward", mutates_args=())
def softmax_rows_backward(p: Tensor, dp: Tensor) -> Tensor:
    p2 = _c(p.reshape(-1, p.shape[-1]))
    dx = torch.empty_like(p2)
    call("isic_softmax_rows_bwd", p2, _c(dp.reshape(p2.shape)), dx, p2.shape[0], p2.shape[1])
    return dx.reshape(p.shape)


@softmax_rows_backward.register_fake
def _(p, dp):
    return p.new_empty(p.shape, dtype=torch.float32)


softmax_rows.register_autograd(lambda ctx, dp: softmax_rows_backward(ctx.saved_tensors[0], dp),
                               setup_context=lambda ctx, inputs, output: ctx.save_for_backward(output))


@custom_op("isic_hip::cross_entropy", mutates_args=())
def cross_entropy(inp: Tensor, labels: Tensor, mode: int) -> Tuple[Tensor, Tensor]:
    """-> (mean loss [], d loss / d inp).  mode 0: logits (`01_train_mil_teacher.py:244`); 1: probabilities through
    log(p + 1e-9) (`05_train_gnns.py:344`)."""
    x = _c(inp.reshape(-1, inp.shape[-1]))
    B, C = x.shape
    loss_ps = torch.empty((B,), device=x.device, dtype=torch.float32)
    loss = torch.empty((1,), device=x.device, dtype=torch.float32)
    d = torch.empty_like(x)
    call("isic_cross_entropy", x, labels.reshape(-1).to(torch.int64).contiguous(), B, C, int(mode), 1.0, loss_ps, loss, d)
    return loss.reshape(()), d.reshape(inp.shape)


@cross_entropy.register_fake
def _(inp, labels, mode):
    return inp.new_empty((), dtype=torch.float32), inp.new_empty(inp.shape, dtype=torch.float32)


def _ce_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.mark_non_differentiable(output[1])


cross_entropy.register_autograd(lambda ctx, dloss, _dd: (ctx.saved_tensors[0] * dloss, None, None), setup_context=_ce_setup)


# ----------------------------------------------------------------------------------------------- dense helpers (no autograd)
@custom_op("isic_hip::gemm_f32", mutates_args=())
def gemm_f32(a: Tensor, b: Tensor, trans_a: bool, trans_b: bool, bias: Optional[Tensor], act: int) -> Tensor:
    return _o.gemm(_c(a), _c(b), trans_a=trans_a, trans_b=trans_b, bias=_c(bias) if bias is not None else None, act=int(act))


@gemm_f32.register_fake
def _(a, b, trans_a, trans_b, bias, act):
    M = a.shape[1] if trans_a else a.shape[0]
    N = b.shape[0] if trans_b else b.shape[1]
    return a.new_empty((M, N), dtype=torch.float32)


def graph_tensors(graph):
    """The six CSR tensors of a ``graph.GraphBatch`` in the order ``isic_hip::spmm`` takes them."""
    return graph.rowptr, graph.col, graph.val, graph.rowptr_t, graph.col_t, graph.val_t


def drop_args(drop):
    """``ops.DropoutSpec`` (or None) -> the four scalars a registered op takes for its dropout site (the step has to be
    folded into ``stream`` by the caller: registered ops take no device step clock)."""
    if drop is None or not drop.active:
        return 0, 1.0, 0, 0
    if drop.clock is not None:
        raise ValueError("registered ops take the dropout stream as a scalar; use a DropoutSpec without a device clock")
    # op schemas take signed 64-bit integers; seeds / streams are uint64 (a seed >= 2^63 is legal: torch.initial_seed() can be):
    # pass the same 64 bits as a signed value -- ctypes turns it back into the unsigned one at the C ABI
    s64 = lambda v: int(v) - (1 << 64) if int(v) >= (1 << 63) else int(v)
    return int(drop.threshold), float(drop.scale), s64(drop.seed & 0xFFFFFFFFFFFFFFFF), s64(drop.stream & 0xFFFFFFFFFFFFFFFF)
