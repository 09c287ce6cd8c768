"""Patch-graph structures and message-passing ops over libisic_hip (k-NN build,
destination-major CSR with GCN normalisation, CSR SpMM with autograd)."""
from __future__ import annotations

import torch

from .bags import BagOffsets
from .lib import IsicHipError, call
from .ops import _acc_target, _chk, _f32c, colsum


def knn_indices(x, offsets: BagOffsets, k, return_dist=False):
    """k nearest neighbours of every node inside its own graph -> LOCAL ids [T, k] (int64),
    ascending by distance (`03_build_graphs.py:46-50`)."""
    _chk(x)
    x = _f32c(x)
    T, D = x.shape
    if T != offsets.total:
        raise ValueError(f"x has {T} rows but offsets cover {offsets.total}")
    idx = torch.empty((T, k), device=x.device, dtype=torch.int64)
    dist = torch.empty((T, k), device=x.device, dtype=torch.float32) if return_dist else None
    ws = torch.empty((T,), device=x.device, dtype=torch.float32)
    call("isic_knn_graph", x, offsets.device, offsets.num_bags, D, int(k), offsets.max_bag, T, idx, dist, ws)
    return (idx, dist) if return_dist else idx


class GraphBatch:
    """Destination-major CSR (+ its transpose) of a batch of graphs with GCN symmetric
    normalisation and self loops (PyG ``gcn_norm`` semantics, `05_train_gnns.py:82`)."""

    MODES = {"gcn": 0, "sum": 1, "mean": 2}

    def __init__(self, edge_index, n_nodes, edge_weight=None, mode="gcn"):
        """mode 'gcn': self loops + symmetric normalisation (GCNConv / GCN2Conv); 'sum': plain neighbour
        sum (GINConv); 'mean': neighbour mean (SAGEConv).  'sum'/'mean' keep the edges as given."""
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be [2, E]")
        if not edge_index.is_cuda:
            raise IsicHipError("GraphBatch needs device tensors (no CPU fallback)")
        ei = edge_index.to(torch.int64).contiguous()
        E, n = int(ei.shape[1]), int(n_nodes)
        dev = ei.device
        ew = _f32c(edge_weight) if edge_weight is not None else None
        self.n_nodes, self.num_edges = n, E
        self.rowptr = torch.empty(n + 1, device=dev, dtype=torch.int32)
        self.col = torch.empty(E + n, device=dev, dtype=torch.int32)
        self.val = torch.empty(E + n, device=dev, dtype=torch.float32)
        self.rowptr_t = torch.empty(n + 1, device=dev, dtype=torch.int32)
        self.col_t = torch.empty(E + n, device=dev, dtype=torch.int32)
        self.val_t = torch.empty(E + n, device=dev, dtype=torch.float32)
        self.perm_t = torch.empty(E + n, device=dev, dtype=torch.int32)   # transposed slot -> CSR slot of the same edge
        nbytes = int(call("isic_gcn_csr_workspace_bytes", n, E))
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        self.mode = mode
        call("isic_gcn_csr_build", ei[0], ei[1], ew, E, n, self.MODES[mode], self.rowptr, self.col, self.val,
             self.rowptr_t, self.col_t, self.val_t, self.perm_t, ws, nbytes)


def _graph_from_parts(n_nodes, num_edges, mode, parts):
    """A ``GraphBatch`` over CSR arrays that already exist on the device (no build launch): used by
    ``train.GraphStore`` to assemble a step's batch out of per-graph pieces built once."""
    if mode not in GraphBatch.MODES:
        raise ValueError(f"unknown graph mode '{mode}'")
    g = GraphBatch.__new__(GraphBatch)
    g.n_nodes, g.num_edges, g.mode = int(n_nodes), int(num_edges), mode
    for k in ("rowptr", "col", "val", "rowptr_t", "col_t", "val_t", "perm_t"):
        t = parts[k]
        want = torch.float32 if k.startswith("val") else torch.int32
        if not t.is_cuda or t.dtype != want:
            raise IsicHipError(f"GraphBatch.from_parts: {k} must be a device {want} tensor")
        setattr(g, k, t.contiguous())
    if g.rowptr.numel() != g.n_nodes + 1 or g.rowptr_t.numel() != g.n_nodes + 1:
        raise ValueError("rowptr must have n_nodes + 1 entries")
    return g


GraphBatch.from_parts = staticmethod(_graph_from_parts)


def _spmm_launch(graph, transposed, x, bias, out, alpha, addend, addend_scale):
    """One aggregation launch (forward CSR or its transpose)."""
    rp, c, v = (graph.rowptr_t, graph.col_t, graph.val_t) if transposed else (graph.rowptr, graph.col, graph.val)
    n, F = x.shape
    call("isic_spmm_csr_f32", rp, c, v, x, bias, out, n, F, alpha, addend, addend_scale)


class SpmmFn(torch.autograd.Function):
    """out = alpha * A^ x (+ bias) (+ addend_scale * addend); backward through the transposed CSR."""

    @staticmethod
    def forward(ctx, x, graph, bias, alpha, addend, addend_scale):
        _chk(x, bias, addend)
        x2 = _f32c(x)
        n, F = x2.shape
        if n != graph.n_nodes:
            raise ValueError(f"x has {n} rows, graph has {graph.n_nodes} nodes")
        out = torch.empty_like(x2)
        _spmm_launch(graph, False, x2, _f32c(bias) if bias is not None else None, out, float(alpha),
                     _f32c(addend) if addend is not None else None, float(addend_scale))
        ctx.graph, ctx.alpha, ctx.addend_scale = graph, float(alpha), float(addend_scale)
        ctx.has_bias, ctx.has_addend = bias is not None, addend is not None
        ctx.bias_param = bias                   # the Parameter itself: fused_grad_accumulation adds into its .grad
        return out

    @staticmethod
    def backward(ctx, dy):
        g = ctx.graph
        dy = _f32c(dy)
        dx = db = da = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(dy)
            _spmm_launch(g, True, dy, None, dx, ctx.alpha, None, 0.0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            tgt = _acc_target(ctx.bias_param)
            if tgt is not None:
                colsum(dy, out=tgt, beta=1.0)
            else:
                db = colsum(dy)
        if ctx.has_addend and ctx.needs_input_grad[4]:
            da = dy * ctx.addend_scale
        return dx, None, db, None, da, None


def spmm(x, graph, bias=None, alpha=1.0, addend=None, addend_scale=0.0):
    return SpmmFn.apply(x, graph, bias, alpha, addend, addend_scale)


class GcnBlockFn(torch.autograd.Function):
    """One residual GCN layer of GraphMIL as ONE autograd node (`05_train_gnns.py:184-199`):

        y = dropout(relu(LayerNorm(A^ (h W^T) + b))) + h

    The kernels are those of ``ops.linear`` -> ``spmm`` -> ``ops.layer_norm``; what the single node buys is the backward: the
    two gradient paths into ``h`` (the residual's dy and the convolution's (A^T d) W) meet in the epilogue of the data-gradient
    GEMM (``gemm(..., addend=dy)``) instead of in an elementwise pass autograd would launch over both [T, F] tensors."""

    @staticmethod
    def forward(ctx, h, weight, bias, gamma, beta, graph, eps, drop, residual):
        from .ops import NO_DROP, gemm
        _chk(h, weight, bias, gamma, beta)
        h2, w = _f32c(h), _f32c(weight)
        n, F = h2.shape[0], w.shape[0]
        if n != graph.n_nodes:
            raise ValueError(f"x has {n} rows, graph has {graph.n_nodes} nodes")
        if residual and w.shape[1] != F:
            raise ValueError("a residual GCN block keeps the feature width")
        drop = drop or NO_DROP
        lin = gemm(h2, w, trans_b=True)
        agg = torch.empty_like(lin)
        _spmm_launch(graph, False, lin, _f32c(bias) if bias is not None else None, agg, 1.0, None, 0.0)
        g_, b_ = _f32c(gamma), _f32c(beta)
        y = torch.empty_like(agg)
        mean = torch.empty((n,), device=h2.device, dtype=torch.float32)
        rstd = torch.empty((n,), device=h2.device, dtype=torch.float32)
        call("isic_layernorm_fwd_clk", agg, g_, b_, h2 if residual else None, y, mean, rstd, n, F, float(eps), 1, drop.threshold,
             drop.scale, drop.seed, drop.stream, drop.clock)
        ctx.graph, ctx.drop, ctx.residual = graph, drop, bool(residual)
        ctx.params = (weight, bias, gamma, beta)
        ctx.save_for_backward(h2, w, agg, g_, b_, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .ops import _workspace, gemm
        h2, w, agg, g_, b_, mean, rstd = ctx.saved_tensors
        weight, bias, gamma, beta = ctx.params
        drop = ctx.drop
        n, F = agg.shape
        dy2 = _f32c(dy)
        # LayerNorm (+ReLU +dropout) backward; dgamma / dbeta accumulate
        d_agg = torch.empty_like(agg)
        tg, tb = _acc_target(gamma), _acc_target(beta)
        fused_ln = tg is not None and tb is not None
        dg = tg if fused_ln else torch.zeros((F,), device=agg.device, dtype=torch.float32)
        db_ln = tb if fused_ln else torch.zeros((F,), device=agg.device, dtype=torch.float32)
        ws = _workspace(call("isic_layernorm_bwd_workspace_bytes", F), agg.device)
        # GCNConv's bias gradient = column sums of d_agg: taken by the LayerNorm backward itself for the vector widths
        dbias, want_db = None, bias is not None and ctx.needs_input_grad[2]
        if want_db and F in (64, 128, 256):
            t = _acc_target(bias)
            if t is None:
                t = dbias = torch.zeros((F,), device=agg.device, dtype=torch.float32)
            call("isic_layernorm_bwd_dxsum_ws", dy2, agg, g_, b_, mean, rstd, d_agg, dg, db_ln, t, n, F, 1, drop.threshold,
                 drop.scale, drop.seed, drop.stream, drop.clock, ws, ws.numel() if ws is not None else 0)
            want_db = False
        else:
            call("isic_layernorm_bwd_ws", dy2, agg, g_, b_, mean, rstd, d_agg, dg, db_ln, n, F, 1, drop.threshold, drop.scale,
                 drop.seed, drop.stream, drop.clock, ws, ws.numel() if ws is not None else 0)
        if want_db:
            t = _acc_target(bias)
            if t is not None:
                colsum(d_agg, out=t, beta=1.0)
            else:
                dbias = colsum(d_agg)
        d_lin = torch.empty_like(d_agg)
        _spmm_launch(ctx.graph, True, d_agg, None, d_lin, 1.0, None, 0.0)
        dW = None
        if ctx.needs_input_grad[1]:
            t = _acc_target(weight)
            if t is not None:
                gemm(d_lin, h2, trans_a=True, out=t, beta=1.0)
            else:
                dW = gemm(d_lin, h2, trans_a=True)
        dh = None
        if ctx.needs_input_grad[0]:
            dh = gemm(d_lin, w, addend=dy2 if ctx.residual else None)      # (A^T d) W + dy: both paths in one epilogue
        return dh, dW, dbias, (None if fused_ln else dg), (None if fused_ln else db_ln), None, None, None, None


def gcn_block(h, weight, bias, gamma, beta, graph, eps=1e-5, drop=None, residual=True):
    return GcnBlockFn.apply(h, weight, bias, gamma, beta, graph, eps, drop, residual)


class L2NormalizeFn(torch.autograd.Function):
    """Row-wise x / max(||x||_2, eps): ``F.normalize`` of SAGEConv(normalize=True)."""

    @staticmethod
    def forward(ctx, x, eps):
        _chk(x)
        x2 = _f32c(x)
        y = torch.empty_like(x2)
        n = torch.empty(x2.shape[0], device=x2.device, dtype=torch.float32)
        call("isic_l2normalize_fwd", x2, y, n, x2.shape[0], x2.shape[1], float(eps))
        ctx.save_for_backward(y, n)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, n = ctx.saved_tensors
        dx = torch.empty_like(y)
        call("isic_l2normalize_bwd", _f32c(dy), y, n, dx, y.shape[0], y.shape[1], ctx.eps)
        return dx, None


def l2_normalize(x, eps=1e-12):
    return L2NormalizeFn.apply(x, eps)


class GatFn(torch.autograd.Function):
    """PyG GATConv message passing on a 'gcn'-mode GraphBatch (self loops re-added): per-destination
    edge softmax of leaky_relu(<x', att_src>[src] + <x', att_dst>[dst]) and the weighted neighbour sum."""

    @staticmethod
    def forward(ctx, xp, att_src, att_dst, bias, graph, heads, slope, drop):
        from .ops import NO_DROP
        _chk(xp, att_src, att_dst, bias)
        xp = _f32c(xp)
        N, HF = xp.shape
        F_ = HF // heads
        a_s, a_d = _f32c(att_src).reshape(heads, F_), _f32c(att_dst).reshape(heads, F_)
        drop = drop or NO_DROP
        dev = xp.device
        al = torch.empty((N, heads), device=dev, dtype=torch.float32)
        ar = torch.empty((N, heads), device=dev, dtype=torch.float32)
        call("isic_gat_scores", xp, a_s, a_d, al, ar, N, heads, F_)
        nnz = graph.col.numel()
        alpha = torch.empty((nnz, heads), device=dev, dtype=torch.float32)
        out = torch.empty_like(xp)
        call("isic_gat_fwd", xp, al, ar, graph.rowptr, graph.col, _f32c(bias) if bias is not None else None, out, alpha, N,
             heads, F_, float(slope), drop.threshold, drop.scale, drop.seed, drop.stream)
        ctx.graph, ctx.cfg = graph, (N, heads, F_, float(slope), drop, bias is not None, att_src.shape)
        ctx.save_for_backward(xp, a_s, a_d, al, ar, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .ops import gemm
        xp, a_s, a_d, al, ar, alpha = ctx.saved_tensors
        N, H, F_, slope, drop, has_bias, att_shape = ctx.cfg
        g = ctx.graph
        dout = _f32c(dout)
        dev = xp.device
        de = torch.empty_like(alpha)
        dar = torch.empty((N, H), device=dev, dtype=torch.float32)
        dal = torch.empty((N, H), device=dev, dtype=torch.float32)
        dxp = torch.empty_like(xp)
        call("isic_gat_bwd", dout, xp, alpha, al, ar, a_s, a_d, g.rowptr, g.col, g.rowptr_t, g.col_t, g.perm_t, de, dar,
             dal, dxp, N, H, F_, slope, drop.threshold, drop.scale, drop.seed, drop.stream)
        # d att_src[h,:] = dal[:,h]^T x'[:,h,:]   (strided views: one small GEMM per head)
        d_as = torch.empty((H, F_), device=dev, dtype=torch.float32)
        d_ad = torch.empty((H, F_), device=dev, dtype=torch.float32)
        for h in range(H):
            xh = xp[:, h * F_:(h + 1) * F_]
            gemm(dal[:, h:h + 1], xh, trans_a=True, out=d_as[h:h + 1])
            gemm(dar[:, h:h + 1], xh, trans_a=True, out=d_ad[h:h + 1])
        db = colsum(dout) if has_bias else None
        return dxp, d_as.reshape(att_shape), d_ad.reshape(att_shape), db, None, None, None, None


def gat_conv(xp, att_src, att_dst, bias, graph, heads, negative_slope=0.2, drop=None):
    return GatFn.apply(xp, att_src, att_dst, bias, graph, heads, negative_slope, drop)


class EdgeAttnFn(torch.autograd.Function):
    """Edge-softmax attention aggregation of ``isic_edge_attn_fwd/bwd``.
    mode 0 (GATv2Conv): ks = x_l (also the values), qd = x_r, att[H,F]; 'gcn'-mode GraphBatch (self loops re-added).
    mode 1 (TransformerConv): ks = key, qd = query, v = value, scale = 1/sqrt(F); 'sum'-mode GraphBatch."""

    @staticmethod
    def forward(ctx, mode, ks, qd, v, att, bias, graph, heads, slope, scale, drop):
        from .ops import NO_DROP
        _chk(ks, qd, v, att, bias)
        ks, qd = _f32c(ks), _f32c(qd)
        v = ks if mode == 0 else _f32c(v)
        N, HF = ks.shape
        F_ = HF // heads
        a = _f32c(att).reshape(heads, F_) if att is not None else None
        drop = drop or NO_DROP
        alpha = torch.empty((graph.col.numel(), heads), device=ks.device, dtype=torch.float32)
        out = torch.empty_like(ks)
        call("isic_edge_attn_fwd", int(mode), ks, qd, v, a, graph.rowptr, graph.col, _f32c(bias) if bias is not None else None,
             out, alpha, N, heads, F_, float(slope), float(scale), drop.threshold, drop.scale, drop.seed, drop.stream)
        ctx.graph, ctx.cfg = graph, (int(mode), N, heads, F_, float(slope), float(scale), drop, bias is not None,
                                     att.shape if att is not None else None)
        ctx.save_for_backward(ks, qd, v, a, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        ks, qd, v, a, alpha = ctx.saved_tensors
        mode, N, H, F_, slope, scale, drop, has_bias, att_shape = ctx.cfg
        g = ctx.graph
        dout = _f32c(dout)
        dev = ks.device
        de = torch.empty_like(alpha)
        dqd, dks = torch.empty_like(qd), torch.empty_like(ks)
        dv = torch.empty_like(v) if mode == 1 else None
        datt = torch.zeros((H, F_), device=dev, dtype=torch.float32) if mode == 0 else None
        call("isic_edge_attn_bwd", mode, dout, ks, qd, v, a, alpha, g.rowptr, g.col, g.rowptr_t, g.col_t, g.perm_t, de, dqd,
             dks, dv, datt, N, H, F_, slope, scale, drop.threshold, drop.scale, drop.seed, drop.stream)
        db = colsum(dout) if has_bias else None
        return (None, dks, dqd, dv, datt.reshape(att_shape) if datt is not None else None, db, None, None, None, None, None)


def gatv2_conv(xl, xr, att, bias, graph, heads, negative_slope=0.2, drop=None):
    return EdgeAttnFn.apply(0, xl, xr, None, att, bias, graph, heads, negative_slope, 1.0, drop)


def transformer_attention(q, k, v, graph, heads, drop=None):
    F_ = q.shape[1] // heads
    return EdgeAttnFn.apply(1, k, q, v, None, None, graph, heads, 0.0, 1.0 / (F_ ** 0.5), drop)


class FaFn(torch.autograd.Function):
    """FAConv propagation on a 'gcn'-mode GraphBatch: out = sum tanh(<x,att_l>[src] + <x,att_r>[dst]) * w * x[src] +
    eps * x0 (``isic_fa_fwd/bwd``; the two scalar scores per node come from ``isic_gat_scores`` with one head)."""

    @staticmethod
    def forward(ctx, x, x0, att_l, att_r, graph, eps, drop):
        from .ops import NO_DROP
        _chk(x, x0, att_l, att_r)
        x, x0 = _f32c(x), _f32c(x0)
        N, F_ = x.shape
        al_w, ar_w = _f32c(att_l).reshape(1, F_), _f32c(att_r).reshape(1, F_)
        drop = drop or NO_DROP
        dev = x.device
        al = torch.empty((N, 1), device=dev, dtype=torch.float32)
        ar = torch.empty((N, 1), device=dev, dtype=torch.float32)
        call("isic_gat_scores", x, al_w, ar_w, al, ar, N, 1, F_)
        coef = torch.empty(graph.col.numel(), device=dev, dtype=torch.float32)
        out = torch.empty_like(x)
        call("isic_fa_fwd", x, x0, al, ar, graph.rowptr, graph.col, graph.val, out, coef, N, F_, float(eps), drop.threshold,
             drop.scale, drop.seed, drop.stream)
        ctx.graph, ctx.cfg = graph, (N, F_, float(eps), drop, att_l.shape, att_r.shape)
        ctx.save_for_backward(x, al_w, ar_w, coef)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .ops import gemm
        x, al_w, ar_w, coef = ctx.saved_tensors
        N, F_, eps, drop, shp_l, shp_r = ctx.cfg
        g = ctx.graph
        dout = _f32c(dout)
        dev = x.device
        de = torch.empty_like(coef)
        dar = torch.empty((N, 1), device=dev, dtype=torch.float32)
        dal = torch.empty((N, 1), device=dev, dtype=torch.float32)
        dx = torch.empty_like(x)
        call("isic_fa_bwd", dout, x, coef, al_w, ar_w, g.rowptr, g.col, g.val, g.rowptr_t, g.col_t, g.val_t, g.perm_t, de, dar,
             dal, dx, N, F_, drop.threshold, drop.scale, drop.seed, drop.stream)
        d_l = torch.empty((1, F_), device=dev, dtype=torch.float32)
        d_r = torch.empty((1, F_), device=dev, dtype=torch.float32)
        gemm(dal, x, trans_a=True, out=d_l)              # d att_l = dal^T x
        gemm(dar, x, trans_a=True, out=d_r)
        return dx, dout * eps, d_l.reshape(shp_l), d_r.reshape(shp_r), None, None, None


def fa_conv(x, x0, att_l, att_r, graph, eps=0.1, drop=None):
    return FaFn.apply(x, x0, att_l, att_r, graph, eps, drop)
