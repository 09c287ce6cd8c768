"""Custom torch ops (``torch.autograd.Function``) over the libisic_hip C ABI.

Every function here launches hand-written HIP kernels on the current torch
stream; torch only owns the device memory and the autograd graph.  CPU tensors
are rejected (no fallback).
"""
from __future__ import annotations

import ctypes
import math

import torch

from .lib import ERR_UNSUPPORTED, IsicHipError, call

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2


# --------------------------------------------------------------------------- dropout bookkeeping
class DropoutSpec:
    """Counter-based dropout site: keep iff philox_word(i; seed, stream) >= threshold."""

    __slots__ = ("p", "threshold", "scale", "seed", "stream", "clock")

    def __init__(self, p=0.0, seed=0, stream=0, clock=None):
        """``clock``: device step clock (uint64[2] tensor, ``graphs.StepClock``): the kernels then add clock[0] * 1024 to
        ``stream`` themselves, so that a captured (hipGraph) train step draws new words on every replay."""
        self.p = float(p)
        if not 0.0 <= self.p < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), got {p}")
        self.threshold = min(int(math.floor(self.p * 4294967296.0)), 0xFFFFFFFF) if self.p > 0 else 0
        # fp32 arithmetic, like the CPU definition: 1f / (1f - p)
        self.scale = float(torch.tensor(1.0, dtype=torch.float32) / (torch.tensor(1.0, dtype=torch.float32)
                                                                   - torch.tensor(self.p, dtype=torch.float32)))
        self.seed = int(seed)
        self.stream = int(stream)
        self.clock = clock

    @property
    def active(self):
        return self.threshold != 0


NO_DROP = DropoutSpec(0.0)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise IsicHipError("isic_hip ops run on the MI355X only: got a CPU tensor (no CPU fallback)")
        if t.dtype not in (torch.float32, torch.int64, torch.bfloat16, torch.int32, torch.float64):
            raise ValueError(f"unsupported dtype {t.dtype}")


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


# --------------------------------------------------------------------------- gradient accumulation inside the kernels
_FUSED_ACC = False


class fused_grad_accumulation:
    """Inside this context the backward of ``linear`` / ``layer_norm`` ADDS a parameter's gradient into its existing
    ``.grad`` buffer in the producing kernel's epilogue (GEMM with beta = 1, the column sums and LayerNorm's dgamma / dbeta
    likewise) and hands autograd ``None`` -- instead of materialising the gradient and letting AccumulateGrad launch one
    tiny add kernel per parameter (~30 per GraphMIL step, each bound by launch latency).  Meant for the train loops,
    where every parameter is a view into the optimizer's flat gradient buffer that ``zero_grad`` has zeroed
    (`05_train_gnns.py:341-346`: zero_grad -> backward -> step); only ``loss.backward()`` semantics are preserved
    (``torch.autograd.grad`` would not see these gradients)."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        global _FUSED_ACC
        self.prev, _FUSED_ACC = _FUSED_ACC, self.enabled
        return self

    def __exit__(self, *exc):
        global _FUSED_ACC
        _FUSED_ACC = self.prev
        return False


def _acc_target(p):
    """The buffer a backward kernel may accumulate into directly, or None."""
    if not _FUSED_ACC or not isinstance(p, torch.nn.Parameter):
        return None
    g = p.grad
    if g is None or not g.is_cuda or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


# --------------------------------------------------------------------------- raw helpers (no autograd)
def gemm(a, b, trans_a=False, trans_b=False, bias=None, act=ACT_NONE, out=None, beta=0.0, addend=None):
    """out[M,N] = act(op(a) @ op(b) + bias) + beta*out (+ addend)   (fp32 MFMA)."""
    M = a.shape[1] if trans_a else a.shape[0]
    K = a.shape[0] if trans_a else a.shape[1]
    N = b.shape[0] if trans_b else b.shape[1]
    Kb = b.shape[1] if trans_b else b.shape[0]
    if K != Kb:
        raise ValueError(f"gemm shape mismatch: op(a) is [{M},{K}], op(b) is [{Kb},{N}]")
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    for t in (a, b, out):      # row-major with a leading dimension: strided row views are fine, the inner stride is not
        if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1) or not t.is_cuda or t.dtype != torch.float32:
            raise IsicHipError("gemm operands must be 2-D fp32 device tensors with unit inner stride")
    ws = _workspace(call("isic_gemm_f32_workspace_bytes", int(trans_a), int(trans_b), M, N, K), a.device)
    if addend is not None:
        if addend.shape != out.shape or addend.dtype != torch.float32 or not addend.is_cuda or (N > 1 and addend.stride(1) != 1):
            raise IsicHipError("gemm addend must be an fp32 device tensor of the output's shape with unit inner stride")
        call("isic_gemm_f32_add_ws", int(trans_a), int(trans_b), M, N, K, a.data_ptr(), max(a.stride(0), a.shape[1]),
             b.data_ptr(), max(b.stride(0), b.shape[1]), out.data_ptr(), max(out.stride(0), out.shape[1]), bias, act,
             float(beta), addend.data_ptr(), max(addend.stride(0), N), ws, ws.numel() if ws is not None else 0)
        return out
    call("isic_gemm_f32_ws", int(trans_a), int(trans_b), M, N, K, a.data_ptr(), max(a.stride(0), a.shape[1]), b.data_ptr(),
         max(b.stride(0), b.shape[1]), out.data_ptr(), max(out.stride(0), out.shape[1]), bias, act, float(beta),
         ws, ws.numel() if ws is not None else 0)
    return out


_WS = {}


def _workspace(nbytes, device):
    """Split-K / row-chunk partials of the deterministic reductions: one buffer per device, grown to the largest request
    (the launches that use it are ordered on the stream; a fixed address keeps a captured graph valid)."""
    if not nbytes:
        return None
    key = (device.type, device.index)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = torch.empty(max(int(nbytes), 64 << 20), device=device, dtype=torch.uint8)
    return ws


def colsum(x, out=None, beta=0.0):
    if out is None:
        out = torch.empty((x.shape[1],), device=x.device, dtype=torch.float32)
    ws = _workspace(call("isic_colsum_f32_workspace_bytes", x.shape[0], x.shape[1]), x.device)
    if x.dim() != 2 or (x.shape[1] > 1 and x.stride(1) != 1) or not x.is_cuda or x.dtype != torch.float32:
        raise IsicHipError("colsum takes a 2-D fp32 device tensor with unit inner stride (a column slice of a matrix is fine)")
    xp = x if x.is_contiguous() else x.data_ptr()        # a column slice: the kernel walks rows by x.stride(0)
    call("isic_colsum_f32_ws", xp, x.shape[0], x.shape[1], x.stride(0), out, float(beta), ws,
         ws.numel() if ws is not None else 0)
    return out


# --------------------------------------------------------------------------- Linear (+ReLU/+tanh, +dropout)
class LinearFn(torch.autograd.Function):
    """y = dropout(act(x W^T + b)).  nn.Linear of utils_g_mil.py:49-63 etc."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, drop):
        _chk(x, weight, bias)
        x2 = _f32c(x.reshape(-1, x.shape[-1]))
        w = _f32c(weight)
        b = _f32c(bias) if bias is not None else None
        drop = drop or NO_DROP
        if drop.active and act != ACT_RELU:
            raise ValueError("fused dropout is defined after ReLU only")
        if drop.active:
            y = gemm(x2, w, trans_b=True, bias=b, act=ACT_NONE)
            call("isic_relu_dropout_fwd_clk_f32", y, y.numel(), drop.threshold, drop.scale, drop.seed, drop.stream, drop.clock)
        else:
            y = gemm(x2, w, trans_b=True, bias=b, act=act)
        ctx.act, ctx.drop_scale, ctx.has_bias = act, (drop.scale if drop.active else 1.0), bias is not None
        ctx.xshape = x.shape
        ctx.params = (weight, bias)              # the Parameters themselves: fused_grad_accumulation adds into their .grad
        ctx.save_for_backward(x2, w, y if act != ACT_NONE else None)
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        g = _f32c(dy.reshape(-1, w.shape[0]))
        if ctx.act == ACT_RELU:
            g2 = torch.empty_like(g)
            call("isic_relu_dropout_bwd_out_f32", y, g, g2, g.numel(), ctx.drop_scale)
            g = g2
        elif ctx.act == ACT_TANH:
            g2 = torch.empty_like(g)
            call("isic_tanh_bwd_f32", g, y, g2, g.numel())
            g = g2
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm(g, w).reshape(ctx.xshape)
        wp, bp = ctx.params
        if ctx.needs_input_grad[1]:
            tgt = _acc_target(wp)
            if tgt is not None:
                gemm(g, x2, trans_a=True, out=tgt, beta=1.0)
            else:
                dw = gemm(g, x2, trans_a=True)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            tgt = _acc_target(bp)
            if tgt is not None:
                colsum(g, out=tgt, beta=1.0)
            else:
                db = colsum(g)
        return dx, dw, db, None, None


class LinearRowsFn(torch.autograd.Function):
    """y = x_store[rows] W^T + b without materialising x_store[rows]: the GEMM reads its row operand through the index (forward:
    A rows; weight gradient dW = dY^T x_store[rows]: the k rows of B) -- `isic_gemm_f32_rows_ws`.  The input projection of a
    batch of graphs straight out of the resident record store (`05_train_gnns.py:340-343` re-uploads the node features of
    every graph at every step; round 2 gathered them with an index_select: 154 MB read + written per step).  ``x_store`` takes
    no gradient.  Shapes the gathering kernel does not take fall back to gather + ``linear``."""

    @staticmethod
    def forward(ctx, x_store, rows, n_rows, weight, bias):
        _chk(x_store, weight, bias)
        xs, w = _f32c(x_store), _f32c(weight)
        b = _f32c(bias) if bias is not None else None
        M, K, N = int(n_rows), xs.shape[1], w.shape[0]
        if rows.dtype != torch.int32 or not rows.is_cuda or rows.numel() < (M + 7) // 8 * 8:
            raise IsicHipError("linear_rows: rows must be a device int32 tensor padded to a multiple of 8 entries")
        y = torch.empty((M, N), device=xs.device, dtype=torch.float32)
        ws = _workspace(call("isic_gemm_f32_workspace_bytes", 0, 1, M, N, K), xs.device)
        try:
            call("isic_gemm_f32_rows_ws", 0, 1, M, N, K, xs, K, rows, w, K, None, y, N, b, ACT_NONE, 0.0, ws,
                 ws.numel() if ws is not None else 0)
            ctx.gathered = None
        except IsicHipError as e:
            if e.code != ERR_UNSUPPORTED:
                raise
            ctx.gathered = xs[rows[:M].long()]
            gemm(ctx.gathered, w, trans_b=True, bias=b, out=y)
        ctx.params, ctx.M = (weight, bias), M
        ctx.save_for_backward(xs, rows)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, rows = ctx.saved_tensors
        weight, bias = ctx.params
        g = _f32c(dy)
        M, N, K = ctx.M, g.shape[1], xs.shape[1]
        dw = db = None
        if ctx.needs_input_grad[3]:
            tgt = _acc_target(weight)
            out = tgt if tgt is not None else torch.empty((N, K), device=g.device, dtype=torch.float32)
            beta = 1.0 if tgt is not None else 0.0
            if ctx.gathered is not None:
                gemm(g, ctx.gathered, trans_a=True, out=out, beta=beta)
            else:
                ws = _workspace(call("isic_gemm_f32_workspace_bytes", 1, 0, N, K, M), g.device)
                try:
                    call("isic_gemm_f32_rows_ws", 1, 0, N, K, M, g, N, None, xs, K, rows, out, K, None, ACT_NONE, beta, ws,
                         ws.numel() if ws is not None else 0)
                except IsicHipError as e:
                    if e.code != ERR_UNSUPPORTED:
                        raise
                    gemm(g, xs[rows[:M].long()], trans_a=True, out=out, beta=beta)
            dw = None if tgt is not None else out
        if bias is not None and ctx.needs_input_grad[4]:
            tgt = _acc_target(bias)
            if tgt is not None:
                colsum(g, out=tgt, beta=1.0)
            else:
                db = colsum(g)
        return None, None, None, dw, db


def linear_rows(x_store, rows, n_rows, weight, bias=None):
    return LinearRowsFn.apply(x_store, rows, n_rows, weight, bias)


def linear(x, weight, bias=None, act=ACT_NONE, drop=None):
    return LinearFn.apply(x, weight, bias, act, drop)


class ReluDropoutFn(torch.autograd.Function):
    """y = dropout(relu(x)) (`05_train_gnns.py:189-190` without LayerNorm)."""

    @staticmethod
    def forward(ctx, x, drop):
        _chk(x)
        drop = drop or NO_DROP
        y = _f32c(x).clone()
        call("isic_relu_dropout_fwd_clk_f32", y, y.numel(), drop.threshold, drop.scale, drop.seed, drop.stream, drop.clock)
        ctx.scale = drop.scale if drop.active else 1.0
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        g = _f32c(dy).clone()
        call("isic_relu_dropout_bwd_f32", y, g, g.numel(), ctx.scale)
        return g, None


def relu_dropout(x, drop=None):
    return ReluDropoutFn.apply(x, drop)


# --------------------------------------------------------------------------- many small tensors, one launch
def multi_copy(dsts, srcs, accumulate=False):
    """dst[i] (+)= src[i] for up to 32 dense fp32 device tensors of equal sizes pairwise, in ONE launch."""
    n = len(dsts)
    if n != len(srcs):
        raise ValueError("multi_copy: as many sources as destinations")
    for lo in range(0, n, 32):
        d, s_ = dsts[lo:lo + 32], srcs[lo:lo + 32]
        for x, y in zip(d, s_):
            _chk(x, y)
            if x.numel() != y.numel() or x.dtype != torch.float32 or y.dtype != torch.float32 or not x.is_contiguous() \
                    or not y.is_contiguous():
                raise IsicHipError("multi_copy takes dense fp32 tensors of pairwise equal sizes")
        m = len(d)
        dp = (ctypes.c_void_p * m)(*[x.data_ptr() for x in d])
        sp = (ctypes.c_void_p * m)(*[y.data_ptr() for y in s_])
        cn = (ctypes.c_int64 * m)(*[x.numel() for x in d])
        call("isic_multi_copy_f32", m, ctypes.addressof(dp), ctypes.addressof(sp), ctypes.addressof(cn), int(accumulate))


class HeadParamsFn(torch.autograd.Function):
    """The per-head attention parameters (`05_train_gnns.py:126-131`: heads x Sequential(Linear(H, A), Tanh, Linear(A, 1)))
    as the fused operands of ``attn_pool``: (W_0, b_0, w_0, c_0, W_1, ...) -> W2[heads*A, H], b2[heads*A], w3[heads, A],
    b3[heads].  One gather launch forward; backward one scatter launch that adds the slices into the parameters' ``.grad``
    (under ``fused_grad_accumulation``) -- instead of four concatenations and 4 * heads AccumulateGrad kernels."""

    @staticmethod
    def forward(ctx, *params):
        _chk(*params)
        heads = len(params) // 4
        A, H = params[0].shape
        nW, nb = heads * A * H, heads * A
        flat = torch.empty(nW + 2 * nb + heads, device=params[0].device, dtype=torch.float32)
        W2, b2 = flat[:nW].view(heads * A, H), flat[nW:nW + nb]
        w3, b3 = flat[nW + nb:nW + 2 * nb].view(heads, A), flat[nW + 2 * nb:]
        dst = []
        for k in range(heads):
            dst += [W2[k * A:(k + 1) * A], b2[k * A:(k + 1) * A], w3[k], b3[k:k + 1]]
        multi_copy(dst, [_f32c(p.detach()) for p in params])
        ctx.params, ctx.dims = params, (heads, A, H)
        return W2, b2, w3, b3

    @staticmethod
    def backward(ctx, gW2, gb2, gw3, gb3):
        heads, A, H = ctx.dims
        gs = []
        for k in range(heads):
            gs += [None if gW2 is None else gW2[k * A:(k + 1) * A], None if gb2 is None else gb2[k * A:(k + 1) * A],
                   None if gw3 is None else gw3.reshape(heads, A)[k], None if gb3 is None else gb3.reshape(heads)[k:k + 1]]
        tg = [_acc_target(p) for p in ctx.params]
        if all(t is not None for t in tg) and all(g is not None and g.is_contiguous() for g in gs):
            multi_copy(tg, gs, accumulate=True)
            return (None,) * len(ctx.params)
        return tuple(None if g is None else g.reshape(p.shape) for g, p in zip(gs, ctx.params))


def head_params(attention_layers):
    """Fused attn_pool operands of a ModuleList of Sequential(Linear, Tanh, Linear) heads."""
    ps = []
    for a in attention_layers:
        ps += [a[0].weight, a[0].bias, a[2].weight, a[2].bias]
    return HeadParamsFn.apply(*ps)


# --------------------------------------------------------------------------- attention pool
class AttnPoolFn(torch.autograd.Function):
    """Attention scores + segmented softmax + pooling over ragged bags.

    t = tanh(h W2^T + b2); s = t w3^T + b3 per head; att = softmax over each bag;
    z = mean_heads sum_n att*h; teacher form (W4 given): patch logits/probs and
    class-space pooled bag logits/probs.  Differentiable outputs: z, bag_logits.
    """

    @staticmethod
    def forward(ctx, h, W2, b2, w3, b3, W4, b4, offsets, max_bag, heads):
        _chk(h, W2, b2, w3, b3, W4, b4, offsets)
        ctx.set_materialize_grads(False)
        h = _f32c(h)
        T, H = h.shape
        A = W2.shape[0] // heads
        B = offsets.numel() - 1
        dev = h.device
        t = gemm(h, _f32c(W2), trans_b=True, bias=_f32c(b2), act=ACT_TANH)
        w3c, b3c = _f32c(w3).reshape(heads, A), _f32c(b3).reshape(heads)
        att = torch.empty((T, heads), device=dev, dtype=torch.float32)
        z = torch.empty((B, H), device=dev, dtype=torch.float32)
        teacher = W4 is not None
        if teacher:
            C = W4.shape[0]
            W4c, b4c = _f32c(W4), _f32c(b4)
            P = torch.empty((T, C), device=dev, dtype=torch.float32)
            PP = torch.empty((T, C), device=dev, dtype=torch.float32)
            BL = torch.empty((B, C), device=dev, dtype=torch.float32)
            BP = torch.empty((B, C), device=dev, dtype=torch.float32)
        else:
            C, W4c, b4c, P, PP, BL, BP = 0, None, None, None, None, None, None
        call("isic_attn_pool_fwd", h, t, w3c, b3c, W4c, b4c, offsets, B, H, A, heads, C, int(max_bag), att, z, P, PP,
             BL, BP)
        ctx.dims = (B, H, A, heads, C, int(max_bag))
        ctx.teacher = teacher
        ctx.params = (W2, b2, w3, b3, W4, b4)          # the callers' tensors: Parameters take their gradients in place
        ctx.save_for_backward(h, t, att, P, _f32c(W2), w3c, W4c, offsets)
        if teacher:
            ctx.mark_non_differentiable(att, P, PP, BP)
            return z, att, P, PP, BL, BP
        ctx.mark_non_differentiable(att)
        return z, att

    @staticmethod
    def backward(ctx, dz, *rest):
        h, t, att, P, W2, w3c, W4c, offsets = ctx.saved_tensors
        B, H, A, heads, C, max_bag = ctx.dims
        dBL = rest[3] if ctx.teacher else None
        T = h.shape[0]
        dev = h.device
        dz = _f32c(dz) if dz is not None else None
        dBL = _f32c(dBL) if dBL is not None else None
        d_h = torch.empty((T, H), device=dev, dtype=torch.float32)
        d_u = torch.empty((T, heads * A), device=dev, dtype=torch.float32)
        d_s = torch.empty((T, heads), device=dev, dtype=torch.float32)
        d_P = torch.empty((T, C), device=dev, dtype=torch.float32) if ctx.teacher else None
        # Parameters whose .grad lives in the flat gradient buffer (optim.FlatParams) are accumulated INTO by the reduction
        # kernels themselves (GEMM beta = 1, column sums with beta = 1): no gradient tensor, no AccumulateGrad add launch
        # (six of them per teacher step: 11 % of a 0.63 ms step at 256 bags)
        pW2, pb2, pw3, pb3, pW4, pb4 = ctx.params
        tW2, tb2, tw3, tb3 = _acc_target(pW2), _acc_target(pb2), _acc_target(pw3), _acc_target(pb3)
        small_fused = tb2 is not None and tw3 is not None and tb3 is not None
        db2 = dw3 = db3 = None
        if H <= 128 and A <= 128:
            # db2 = sum_n d_u, dw3[k, j] = sum_n d_s[n,k] t[n, kA+j], db3 = sum_n d_s: per-bag sums out of the pool kernel
            # (d_u and t are in its registers), then ONE column sum over the bags
            nA = heads * A
            psum = torch.empty((B, 2 * nA + heads), device=dev, dtype=torch.float32)
            call("isic_attn_pool_bwd_sums", h, t, att, P, w3c, W4c, offsets, B, H, A, heads, C, max_bag, dBL, dz, d_h, 0,
                 d_u, d_s, d_P, psum)
            if small_fused:
                colsum(psum[:, :nA], out=tb2.view(-1), beta=1.0)
                colsum(psum[:, nA:2 * nA], out=tw3.view(-1), beta=1.0)
                colsum(psum[:, 2 * nA:], out=tb3.view(-1), beta=1.0)
            else:
                sums = colsum(psum)
                db2, dw3, db3 = sums[:nA], sums[nA:2 * nA].view(heads, A), sums[2 * nA:]
        else:
            call("isic_attn_pool_bwd", h, t, att, P, w3c, W4c, offsets, B, H, A, heads, C, max_bag, dBL, dz, d_h, 0, d_u,
                 d_s, d_P)
            # dw3[k, j] = sum_n d_s[n,k] * t[n, k*A + j]
            if small_fused and heads == 1:
                colsum(d_u, out=tb2.view(-1), beta=1.0)
                gemm(d_s, t, trans_a=True, out=tw3.view(1, A), beta=1.0)
                colsum(d_s, out=tb3.view(-1), beta=1.0)
            else:
                small_fused = False
                db2 = colsum(d_u)
                if heads == 1:
                    dw3 = gemm(d_s, t, trans_a=True)                 # [1, A]
                else:
                    full = gemm(d_s, t, trans_a=True)                # [heads, heads*A]
                    dw3 = torch.stack([full[k, k * A:(k + 1) * A] for k in range(heads)])
                db3 = colsum(d_s)
        # weight gradient of the first Linear: a GEMM over the T instances
        if tW2 is not None:
            gemm(d_u, h, trans_a=True, out=tW2, beta=1.0)
            dW2 = None
        else:
            dW2 = gemm(d_u, h, trans_a=True)                 # [heads*A, H]
        gemm(d_u, W2, out=d_h, beta=1.0)                     # d_h += d_u W2
        dW4 = db4 = None
        if ctx.teacher:
            tW4, tb4 = _acc_target(pW4), _acc_target(pb4)
            if tW4 is not None:
                gemm(d_P, h, trans_a=True, out=tW4, beta=1.0)
            else:
                dW4 = gemm(d_P, h, trans_a=True)
            if tb4 is not None:
                colsum(d_P, out=tb4, beta=1.0)
            else:
                db4 = colsum(d_P)
        return d_h, dW2, db2, dw3, db3, dW4, db4, None, None, None


def attn_pool(h, W2, b2, w3, b3, offsets, max_bag, heads=1, W4=None, b4=None):
    return AttnPoolFn.apply(h, W2, b2, w3, b3, W4, b4, offsets, max_bag, heads)


# --------------------------------------------------------------------------- LayerNorm (+ReLU +dropout +residual)
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, eps, relu, drop):
        _chk(x, gamma, beta, residual)
        x2 = _f32c(x.reshape(-1, x.shape[-1]))
        M, N = x2.shape
        drop = drop or NO_DROP
        g, b = _f32c(gamma), _f32c(beta)
        res = _f32c(residual.reshape(-1, N)) if residual is not None else None
        y = torch.empty_like(x2)
        mean = torch.empty((M,), device=x2.device, dtype=torch.float32)
        rstd = torch.empty((M,), device=x2.device, dtype=torch.float32)
        call("isic_layernorm_fwd_clk", x2, g, b, res, y, mean, rstd, M, N, float(eps), int(relu), drop.threshold,
             drop.scale, drop.seed, drop.stream, drop.clock)
        ctx.cfg = (M, N, int(relu), drop)
        ctx.has_res = residual is not None
        ctx.xshape = x.shape
        ctx.params = (gamma, beta)
        ctx.save_for_backward(x2, g, b, mean, rstd)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, g, b, mean, rstd = ctx.saved_tensors
        M, N, relu, drop = ctx.cfg
        dy2 = _f32c(dy.reshape(M, N))
        dx = torch.empty_like(x2)
        tg, tb = _acc_target(ctx.params[0]), _acc_target(ctx.params[1])
        fused = tg is not None and tb is not None      # the kernel ACCUMULATES (+=) into dgamma / dbeta
        dg = tg if fused else torch.zeros((N,), device=x2.device, dtype=torch.float32)
        db = tb if fused else torch.zeros((N,), device=x2.device, dtype=torch.float32)
        ws = _workspace(call("isic_layernorm_bwd_workspace_bytes", N), x2.device)
        call("isic_layernorm_bwd_ws", dy2, x2, g, b, mean, rstd, dx, dg, db, M, N, relu, drop.threshold, drop.scale,
             drop.seed, drop.stream, drop.clock, ws, ws.numel() if ws is not None else 0)
        return dx.reshape(ctx.xshape), (None if fused else dg), (None if fused else db), (dy if ctx.has_res else None), None, None, None


def layer_norm(x, gamma, beta, eps=1e-5, relu=False, drop=None, residual=None):
    return LayerNormFn.apply(x, gamma, beta, residual, eps, relu, drop)


# --------------------------------------------------------------------------- softmax / losses
class SoftmaxRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x2 = _f32c(x.reshape(-1, x.shape[-1]))
        p = torch.empty_like(x2)
        call("isic_softmax_rows_fwd", x2, p, x2.shape[0], x2.shape[1])
        ctx.save_for_backward(p)
        ctx.xshape = x.shape
        return p.reshape(x.shape)

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        dp2 = _f32c(dp.reshape(p.shape))
        dx = torch.empty_like(p)
        call("isic_softmax_rows_bwd", p, dp2, dx, p.shape[0], p.shape[1])
        return dx.reshape(ctx.xshape)


def softmax_rows(x):
    return SoftmaxRowsFn.apply(x)


class CrossEntropyFn(torch.autograd.Function):
    """mean_b CE.  mode 0: logits (01:143,244).  mode 1: probabilities through
    log(p + 1e-9) (05:344)."""

    @staticmethod
    def forward(ctx, inp, labels, mode):
        _chk(inp, labels)
        x = _f32c(inp.reshape(-1, inp.shape[-1]))
        B, C = x.shape
        lab = labels.reshape(-1).to(torch.int64).contiguous()
        loss_ps = torch.empty((B,), device=x.device, dtype=torch.float32)
        loss = torch.empty((1,), device=x.device, dtype=torch.float32)
        d = torch.empty_like(x)
        call("isic_cross_entropy", x, lab, B, C, int(mode), 1.0, loss_ps, loss, d)
        ctx.save_for_backward(d)
        ctx.xshape = inp.shape
        ctx.mark_non_differentiable(loss_ps)
        ctx.set_materialize_grads(False)         # (no zero tensor -- a fill launch -- for the unused per-sample output)
        return loss.reshape(()), loss_ps

    @staticmethod
    def backward(ctx, dloss, _dps):
        (d,) = ctx.saved_tensors
        if dloss is None:
            return None, None, None
        if _is_unit_grad(dloss):                 # ops.backward(loss): d loss / d loss = 1 by construction, no multiply launch
            return d.reshape(ctx.xshape), None, None
        return (d * dloss).reshape(ctx.xshape), None, None


# --------------------------------------------------------------------------- GraphMIL head + loss in two launches
_HEAD_COUNTER = {}


class GraphHeadLossFn(torch.autograd.Function):
    """probs, loss = softmax(Linear(dropout(relu(Linear(z))))), mean CE(log(probs + 1e-9), labels): the classifier_light
    head of GraphMIL and the loss of `05_train_gnns.py:344` as ONE autograd node.  The forward launch also computes every
    gradient (for d loss = 1): dz and the blocks' contributions to the parameter gradients; the backward is one more launch
    that adds the contributions into the parameters' gradients (times d loss).  ``probs`` is an output for the caller's
    metrics: it takes no gradient."""

    @staticmethod
    def forward(ctx, z, W1, b1, W2, b2, labels, drop):
        _chk(z, W1, b1, W2, b2, labels)
        z2 = _f32c(z)
        B, H = z2.shape
        D, C = W1.shape[0], W2.shape[0]
        drop = drop or NO_DROP
        dev = z2.device
        # one ticket counter per (device, stream): two head launches in flight on different streams never share one
        key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
        cnt = _HEAD_COUNTER.get(key)
        if cnt is None:
            cnt = _HEAD_COUNTER[key] = torch.zeros(64, device=dev, dtype=torch.int32)
        probs = torch.empty((B, C), device=dev, dtype=torch.float32)
        loss_ps = torch.empty((B,), device=dev, dtype=torch.float32)
        loss = torch.empty((1,), device=dev, dtype=torch.float32)
        dz = torch.empty_like(z2)
        nbytes = int(call("isic_graph_head_workspace_bytes", B, H, D, C))
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)       # lives until the backward: not the shared workspace
        call("isic_graph_head_fwd_bwd", z2, _f32c(W1), _f32c(b1), _f32c(W2), _f32c(b2),
             labels.reshape(-1).to(torch.int64).contiguous(), B, H, D, C, drop.threshold, drop.scale, drop.seed, drop.stream,
             drop.clock, probs, loss_ps, loss, dz, ws, nbytes, cnt)
        ctx.params, ctx.dims, ctx.zshape = (W1, b1, W2, b2), (B, H, D, C), z.shape
        ctx.save_for_backward(dz, ws)
        ctx.mark_non_differentiable(probs, loss_ps)
        ctx.set_materialize_grads(False)
        return probs, loss.reshape(()), loss_ps

    @staticmethod
    def backward(ctx, _dprobs, dloss, _dps):
        if dloss is None:
            return (None,) * 7
        dz, ws = ctx.saved_tensors
        B, H, D, C = ctx.dims
        W1, b1, W2, b2 = ctx.params
        unit = _is_unit_grad(dloss)
        gs = None if unit else _f32c(dloss).reshape(1)
        tg = [_acc_target(p) for p in ctx.params]
        fused = all(t is not None for t in tg)
        outs = tg if fused else [torch.empty(p.shape, device=dz.device, dtype=torch.float32) for p in ctx.params]
        call("isic_graph_head_param_grads", ws, B, H, D, C, gs, outs[0], outs[1], outs[2], outs[3], int(fused))
        gz = (dz if unit else dz * dloss).reshape(ctx.zshape) if ctx.needs_input_grad[0] else None
        return (gz,) + ((None,) * 4 if fused else tuple(outs)) + (None, None)


def graph_head_supported(H, D, C):
    """Whether ``graph_head_loss`` handles Linear(H, D) -> Linear(D, C) (LDS bound, C < 16); else use the operator chain."""
    try:
        call("isic_graph_head_supported", int(H), int(D), int(C))
        return True
    except IsicHipError as e:
        if e.code != ERR_UNSUPPORTED:
            raise
        return False


def graph_head_loss(z, W1, b1, W2, b2, labels, drop=None):
    """-> (probs[B, C], mean loss).  A label outside [0, C) makes the loss NaN."""
    probs, loss, _ = GraphHeadLossFn.apply(z, W1, b1, W2, b2, labels, drop)
    return probs, loss


_UNIT = {}


def _unit_grad(device):
    key = (device.type, device.index)
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), device=device, dtype=torch.float32)
    return t


def _is_unit_grad(t):
    u = _UNIT.get((t.device.type, t.device.index))
    return u is not None and t.data_ptr() == u.data_ptr()


def backward(loss):
    """``loss.backward()`` for a scalar loss of this package without the two launches torch spends on d loss / d loss: the
    seed gradient is a cached device scalar 1 (no fill), and the loss functions recognise it and skip their multiply."""
    torch.autograd.backward(loss, grad_tensors=(_unit_grad(loss.device),))


def cross_entropy(logits, labels):
    return CrossEntropyFn.apply(logits, labels, 0)[0]


def cross_entropy_from_probs(probs, labels):
    return CrossEntropyFn.apply(probs, labels, 1)[0]
