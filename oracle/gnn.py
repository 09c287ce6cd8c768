"""Patch-graph GNN (GraphMIL), restated on torch-CPU (oracle / test infrastructure).

Follows ``GraphMIL`` in `05_train_gnns.py:51-219` and the graph train step of
`05_train_gnns.py:336-346`.

PARITY UNPINNED for the message-passing layers: the reference calls
``torch_geometric.nn.GCNConv`` / ``GCN2Conv`` (`05_train_gnns.py:82,109`) and
``torch_geometric`` is neither vendored nor version-pinned, and is absent here.
``gcn_conv`` / ``gcn2_conv`` restate PyG's published layer definitions:
  GCNConv : x' = x W^T (no bias in lin); add *remaining* self loops (w=1);
            deg[i] = sum_{e: dst=i} w_e ; w^ = deg^-1/2[src] * w * deg^-1/2[dst];
            out[i] = sum_{e: dst=i} w^_e x'[src_e] ; out += bias
  GCN2Conv: beta = log(theta/layer + 1); h = (1-alpha) * A^ x + alpha * x_0;
            out = (1-beta) * h + beta * (h @ weight1)
  GATConv : edge softmax of leaky_relu(<x',a_src>[src] + <x',a_dst>[dst], 0.2) per destination and head,
            self loops re-added, dropout on alpha, concat heads, + bias
  SAGEConv: lin_l(mean_j x_j) + lin_r(x_i), L2-normalised rows (aggr='mean', normalize=True)
  GINConv : nn((1 + eps) x_i + sum_j x_j), eps trainable, init 0
  GATv2Conv: x_l = lin_l(x), x_r = lin_r(x) (both with bias), viewed [N,H,F]; self loops removed then one added per
            node; e = sum_f att[h,f] * leaky_relu(x_l[src] + x_r[dst], 0.2); alpha = softmax over the edges into dst;
            dropout on alpha; out[dst] = sum alpha x_l[src]; concat heads; + bias.  (The reference does NOT widen
            out_dim by the head count for gatv2 although concat=True, `05_train_gnns.py:99-101` vs `:83-86`, so its
            LayerNorm(out_dim) cannot accept the layer's output for heads > 1; restated WITH the widening, like gat.)
  TransformerConv(beta=True): q = lin_query(x), k = lin_key(x), v = lin_value(x) viewed [N,H,F]; alpha = softmax over
            the edges into dst of <q[dst], k[src]> / sqrt(F) (NO self loops added); dropout on alpha;
            out = concat_h sum alpha v[src]; x_r = lin_skip(x); b = sigmoid(lin_beta([out, x_r, out - x_r]));
            out = b * x_r + (1 - b) * out.
  FAConv  : (x, x_0, edge_index) -- the reference passes (h, edge_index, edge_weight), `05_train_gnns.py:184-185`,
            which is not the layer's signature; restated from the published layer with x_0 = the projected input
            (as GCN2Conv): w = gcn_norm(edge_index) with self loops; a = tanh(att_l(x)[src] + att_r(x)[dst]); dropout
            on a; out[dst] = sum a * w * x[src] + eps * x_0[dst] (eps = 0.1).
The ``mlp`` graph model is pure torch in the reference and IS pinned by
``tests/golden/graphmil_mlp_*.npz``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import mil as _mil


def add_remaining_self_loops(edge_index, edge_weight, n):
    """Drop existing self loops, append one loop per node; a node's existing
    loop weight is kept, otherwise 1 (PyG ``add_remaining_self_loops``)."""
    row, col = edge_index[0], edge_index[1]
    keep = row != col
    loop_w = torch.ones(n, dtype=edge_weight.dtype)
    if (~keep).any():
        loop_w[row[~keep]] = edge_weight[~keep]
    loops = torch.arange(n, dtype=edge_index.dtype)
    ei = torch.cat([edge_index[:, keep], torch.stack([loops, loops])], dim=1)
    ew = torch.cat([edge_weight[keep], loop_w])
    return ei, ew


def gcn_norm(edge_index, edge_weight, n, dtype=torch.float32):
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype)
    ei, ew = add_remaining_self_loops(edge_index, edge_weight, n)
    row, col = ei[0], ei[1]
    deg = torch.zeros(n, dtype=ew.dtype).index_add_(0, col, ew)
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0.0
    return ei, dis[row] * ew * dis[col]


def propagate(x, ei, w):
    """out[dst] += w * x[src]  (flow = source_to_target)."""
    out = torch.zeros_like(x)
    return out.index_add_(0, ei[1], x[ei[0]] * w.unsqueeze(1))


def gcn_conv(x, edge_index, edge_weight, lin_weight, bias):
    ei, w = gcn_norm(edge_index, edge_weight, x.size(0), x.dtype)
    return propagate(F.linear(x, lin_weight), ei, w) + bias


def gcn2_conv(x, x0, edge_index, edge_weight, weight1, alpha, theta, layer):
    beta = math.log(theta / layer + 1.0)
    ei, w = gcn_norm(edge_index, edge_weight, x.size(0), x.dtype)
    h = propagate(x, ei, w) * (1.0 - alpha) + alpha * x0
    return (1.0 - beta) * h + beta * (h @ weight1)


def gat_conv(x, edge_index, lin_w, att_src, att_dst, bias, heads, negative_slope=0.2, drop=None):
    """PyG GATConv(in, F, heads, concat=True): x' = lin(x).view(N,H,F); self loops removed then one
    added per node; e = leaky_relu(<x',att_src>[src] + <x',att_dst>[dst]); alpha = softmax over the
    edges into dst; (counter-based) dropout on alpha, element index = csr_slot*H + h with the
    destination-major slot order of the product (edges in edge_index order per row, self loop last);
    out[dst] = sum alpha x'[src]; concat heads; + bias."""
    n = x.size(0)
    xp = F.linear(x, lin_w).view(n, heads, -1)
    al = (xp * att_src.view(1, heads, -1)).sum(-1)
    ar = (xp * att_dst.view(1, heads, -1)).sum(-1)
    keep = edge_index[0] != edge_index[1]
    loops = torch.arange(n, dtype=edge_index.dtype)
    src = torch.cat([edge_index[0][keep], loops])
    dst = torch.cat([edge_index[1][keep], loops])
    # destination-major slot of every edge (stable sort by dst keeps edge order, loop last)
    order = torch.argsort(dst, stable=True)
    slot = torch.empty_like(order)
    slot[order] = torch.arange(order.numel())
    e = F.leaky_relu(al[src] + ar[dst], negative_slope)                       # [E', H]
    emax = torch.full((n, heads), -float("inf"), dtype=x.dtype).scatter_reduce(0, dst.view(-1, 1).expand(-1, heads), e, "amax")
    ex = torch.exp(e - emax[dst])
    den = torch.zeros(n, heads, dtype=x.dtype).index_add_(0, dst, ex)
    alpha = ex / den[dst]
    if drop is not None and drop["p"] > 0.0:
        from . import philox
        nn_ = alpha.shape[0] * heads
        keep_all = torch.from_numpy(philox.dropout_keep(nn_, drop["p"], drop["seed"], drop["stream"])).view(-1, heads)
        scale = torch.tensor(float(philox.dropout_scale(drop["p"])), dtype=x.dtype)
        alpha = torch.where(keep_all[slot], alpha * scale, torch.zeros((), dtype=x.dtype))
    out = torch.zeros(n, heads, xp.shape[-1], dtype=x.dtype).index_add_(0, dst, alpha.unsqueeze(-1) * xp[src])
    return out.reshape(n, -1) + bias


def sage_conv(x, edge_index, lin_l_w, lin_l_b, lin_r_w):
    """PyG SAGEConv(aggr='mean', normalize=True): lin_l(mean_{j in N(i)} x_j) + lin_r(x_i), then
    F.normalize(p=2, dim=-1).  No self loops are added; a node without in-edges aggregates 0."""
    n = x.size(0)
    agg = torch.zeros_like(x).index_add_(0, edge_index[1], x[edge_index[0]])
    deg = torch.zeros(n, dtype=x.dtype).index_add_(0, edge_index[1], torch.ones(edge_index.size(1), dtype=x.dtype))
    agg = agg / deg.clamp(min=1.0).unsqueeze(1)
    out = F.linear(agg, lin_l_w, lin_l_b) + F.linear(x, lin_r_w)
    return F.normalize(out, p=2.0, dim=-1)


def gin_conv(x, edge_index, eps, w0, b0, w2, b2):
    """PyG GINConv(nn, train_eps=True) with nn = Linear-ReLU-Linear (`05_train_gnns.py:89-93`):
    nn((1 + eps) * x_i + sum_{j in N(i)} x_j)."""
    agg = torch.zeros_like(x).index_add_(0, edge_index[1], x[edge_index[0]])
    z = agg + (1.0 + eps) * x
    return F.linear(F.relu(F.linear(z, w0, b0)), w2, b2)


def _dst_major_slots(src, dst):
    """CSR slot of every edge in the destination-major order of the product (stable sort by destination)."""
    order = torch.argsort(dst, stable=True)
    slot = torch.empty_like(order)
    slot[order] = torch.arange(order.numel())
    return slot


def _edge_softmax(e, dst, n):
    heads = e.shape[1]
    emax = torch.full((n, heads), -float("inf"), dtype=e.dtype).scatter_reduce(0, dst.view(-1, 1).expand(-1, heads), e, "amax")
    ex = torch.exp(e - emax[dst])
    den = torch.zeros(n, heads, dtype=e.dtype).index_add_(0, dst, ex)
    return ex / den[dst]


def _alpha_dropout(alpha, slot, drop):
    if drop is None or drop["p"] <= 0.0:
        return alpha
    from . import philox
    heads = alpha.shape[1] if alpha.dim() > 1 else 1
    keep = torch.from_numpy(philox.dropout_keep(alpha.shape[0] * heads, drop["p"], drop["seed"], drop["stream"]))
    keep = keep.view(-1, heads) if alpha.dim() > 1 else keep.view(-1)
    scale = torch.tensor(float(philox.dropout_scale(drop["p"])), dtype=alpha.dtype)
    return torch.where(keep[slot], alpha * scale, torch.zeros((), dtype=alpha.dtype))


def gatv2_conv(x, edge_index, lin_l_w, lin_l_b, lin_r_w, lin_r_b, att, bias, heads, negative_slope=0.2, drop=None):
    """PyG GATv2Conv(in, F, heads, concat=True, share_weights=False) -- see the module docstring."""
    n = x.size(0)
    xl = F.linear(x, lin_l_w, lin_l_b).view(n, heads, -1)
    xr = F.linear(x, lin_r_w, lin_r_b).view(n, heads, -1)
    keep = edge_index[0] != edge_index[1]
    loops = torch.arange(n, dtype=edge_index.dtype)
    src = torch.cat([edge_index[0][keep], loops])
    dst = torch.cat([edge_index[1][keep], loops])
    e = (F.leaky_relu(xl[src] + xr[dst], negative_slope) * att.view(1, heads, -1)).sum(-1)      # [E', H]
    alpha = _alpha_dropout(_edge_softmax(e, dst, n), _dst_major_slots(src, dst), drop)
    out = torch.zeros(n, heads, xl.shape[-1], dtype=x.dtype).index_add_(0, dst, alpha.unsqueeze(-1) * xl[src])
    return out.reshape(n, -1) + bias


def transformer_conv(x, edge_index, p, prefix, heads, drop=None):
    """PyG TransformerConv(in, F, heads, concat=True, beta=True, root_weight=True) -- see the module docstring."""
    n = x.size(0)
    lin = lambda name: F.linear(x, p[f"{prefix}.{name}.weight"], p[f"{prefix}.{name}.bias"])
    q, k, v = (lin(nm).view(n, heads, -1) for nm in ("lin_query", "lin_key", "lin_value"))
    src, dst = edge_index[0], edge_index[1]
    e = (q[dst] * k[src]).sum(-1) / math.sqrt(q.shape[-1])
    alpha = _alpha_dropout(_edge_softmax(e, dst, n), _dst_major_slots(src, dst), drop) if src.numel() else e
    out = torch.zeros(n, heads, v.shape[-1], dtype=x.dtype).index_add_(0, dst, alpha.unsqueeze(-1) * v[src]).reshape(n, -1)
    xr = lin("lin_skip")
    beta = torch.sigmoid(F.linear(torch.cat([out, xr, out - xr], dim=-1), p[f"{prefix}.lin_beta.weight"]))
    return beta * xr + (1.0 - beta) * out


def fa_conv(x, x0, edge_index, att_l_w, att_r_w, eps=0.1, drop=None):
    """PyG FAConv(channels, eps=0.1, normalize=True, add_self_loops=True) -- see the module docstring."""
    n = x.size(0)
    ei, w = gcn_norm(edge_index, None, n, x.dtype)
    src, dst = ei[0], ei[1]
    al = F.linear(x, att_l_w).view(-1)
    ar = F.linear(x, att_r_w).view(-1)
    a = _alpha_dropout(torch.tanh(al[src] + ar[dst]), _dst_major_slots(src, dst), drop)
    out = torch.zeros_like(x).index_add_(0, dst, (a * w).unsqueeze(-1) * x[src])
    return out + eps * x0


DEFAULT_CFG = dict(  # the 05 call site, `05_train_gnns.py:310-326`
    gnn_type="gcn", gnn_hidden=128, gnn_layers=2, gnn_dropout=0.5, gnn_heads=4,
    gnn_concat=True, gcnii_alpha=0.1, gcnii_theta=0.5, att_dim=128, att_heads=4,
    pool_dropout=0.2, classifier_dim=128, classifier_light=True, num_classes=7,
    use_residual=True, use_layer_norm=True,
)


def graphmil_shapes(input_dim, cfg):
    """Parameter names/shapes as ``GraphMIL.__init__`` creates them
    (`05_train_gnns.py:52-154`), for the gnn types the oracle restates."""
    from collections import OrderedDict
    c = dict(DEFAULT_CFG, **cfg)
    F_, t = c["gnn_hidden"], c["gnn_type"]
    s = OrderedDict()
    has_proj = (c["use_residual"] or t in ("fagcn", "gcnii")) and input_dim != F_
    if has_proj:
        s["input_proj.weight"] = (F_, input_dim)
        s["input_proj.bias"] = (F_,)
    in_dim = F_ if has_proj else input_dim
    for i in range(c["gnn_layers"]):
        if t == "gcn":  # PyG GCNConv: bias, lin.weight
            s[f"gnn_layers.{i}.bias"] = (F_,)
            s[f"gnn_layers.{i}.lin.weight"] = (F_, in_dim)
        elif t == "gcnii":
            s[f"gnn_layers.{i}.weight1"] = (F_, F_)
        elif t == "gat":
            hh = c["gnn_heads"]
            s[f"gnn_layers.{i}.att_src"] = (1, hh, F_)
            s[f"gnn_layers.{i}.att_dst"] = (1, hh, F_)
            s[f"gnn_layers.{i}.bias"] = (hh * F_,)
            s[f"gnn_layers.{i}.lin.weight"] = (hh * F_, in_dim)
            in_dim = hh * F_
            continue
        elif t == "graphsage":
            s[f"gnn_layers.{i}.lin_l.weight"] = (F_, in_dim)
            s[f"gnn_layers.{i}.lin_l.bias"] = (F_,)
            s[f"gnn_layers.{i}.lin_r.weight"] = (F_, in_dim)
        elif t == "gin":
            s[f"gnn_layers.{i}.eps"] = (1,)
            s[f"gnn_layers.{i}.nn.0.weight"] = (F_, in_dim)
            s[f"gnn_layers.{i}.nn.0.bias"] = (F_,)
            s[f"gnn_layers.{i}.nn.2.weight"] = (F_, F_)
            s[f"gnn_layers.{i}.nn.2.bias"] = (F_,)
        elif t == "mlp":
            s[f"gnn_layers.{i}.0.weight"] = (F_, in_dim)
            s[f"gnn_layers.{i}.0.bias"] = (F_,)
        elif t == "gatv2":       # PyG GATv2Conv: att, bias, lin_l.{weight,bias}, lin_r.{weight,bias}
            hh = c["gnn_heads"]
            s[f"gnn_layers.{i}.att"] = (1, hh, F_)
            s[f"gnn_layers.{i}.bias"] = (hh * F_,)
            for nm in ("lin_l", "lin_r"):
                s[f"gnn_layers.{i}.{nm}.weight"] = (hh * F_, in_dim)
                s[f"gnn_layers.{i}.{nm}.bias"] = (hh * F_,)
            in_dim = hh * F_
            continue
        elif t == "transformer":  # PyG TransformerConv(beta=True)
            hh = c["gnn_heads"]
            for nm in ("lin_key", "lin_query", "lin_value", "lin_skip"):
                s[f"gnn_layers.{i}.{nm}.weight"] = (hh * F_, in_dim)
                s[f"gnn_layers.{i}.{nm}.bias"] = (hh * F_,)
            s[f"gnn_layers.{i}.lin_beta.weight"] = (1, 3 * hh * F_)
            in_dim = hh * F_
            continue
        elif t == "fagcn":        # PyG FAConv: att_l.weight, att_r.weight
            s[f"gnn_layers.{i}.att_l.weight"] = (1, F_)
            s[f"gnn_layers.{i}.att_r.weight"] = (1, F_)
        else:
            raise ValueError(f"Unsupported gnn_type: {t}")
        in_dim = F_
    Fo = F_ * c["gnn_heads"] if t in ("gat", "gatv2", "transformer") else F_   # concat=True widens the features, 05:86,98
    if c["use_layer_norm"]:
        for i in range(c["gnn_layers"]):
            s[f"layer_norms.{i}.weight"] = (Fo,)
            s[f"layer_norms.{i}.bias"] = (Fo,)
    for h in range(c["att_heads"]):
        s[f"attention_layers.{h}.0.weight"] = (c["att_dim"], Fo)
        s[f"attention_layers.{h}.0.bias"] = (c["att_dim"],)
        s[f"attention_layers.{h}.2.weight"] = (1, c["att_dim"])
        s[f"attention_layers.{h}.2.bias"] = (1,)
    cd = c["classifier_dim"]
    if c["classifier_light"]:
        s["classifier.0.weight"] = (cd, Fo)
        s["classifier.0.bias"] = (cd,)
        s["classifier.3.weight"] = (c["num_classes"], cd)
        s["classifier.3.bias"] = (c["num_classes"],)
    else:
        raise ValueError("oracle restates the light classifier only (05 call site, 05:321)")
    return s


def graphmil_forward(p, cfg, x, edge_index=None, edge_weight=None, drop=None):
    """`05_train_gnns.py:156-219`.  Returns dict(probs, att, hs, z, logits).

    ``drop`` = dict(seed, stream_base) enables counter-based dropout in train
    mode: GNN layer i uses stream ``stream_base + i`` with ``gnn_dropout``; the
    classifier uses ``stream_base + 64`` with ``pool_dropout``.  Optional ``node_offset`` / ``graph_index``: the graph is
    graph ``graph_index`` of a batch and its first node is row ``node_offset`` of the batched node tensor -- its dropout
    words are then those of its rows in the batched tensors (node features [sum N, F], classifier hidden [G, Dc]).
    """
    c = dict(DEFAULT_CFG, **cfg)
    t = c["gnn_type"]
    if "input_proj.weight" in p:                       # 05:168-171
        x_in = F.linear(x, p["input_proj.weight"], p["input_proj.bias"])
    else:
        x_in = x
    h, x0, hs = x_in, x_in, []
    for i in range(c["gnn_layers"]):                   # 05:177-199
        h_prev = h
        if t == "mlp":
            h = F.linear(h, p[f"gnn_layers.{i}.0.weight"], p[f"gnn_layers.{i}.0.bias"])
        elif t == "gcn":
            h = gcn_conv(h, edge_index, edge_weight, p[f"gnn_layers.{i}.lin.weight"], p[f"gnn_layers.{i}.bias"])
        elif t == "gcnii":
            h = gcn2_conv(h, x0, edge_index, edge_weight, p[f"gnn_layers.{i}.weight1"],
                          c["gcnii_alpha"], c["gcnii_theta"], i + 1)
        elif t == "gat":
            adrop = None
            if drop is not None and c["gnn_dropout"] > 0:
                adrop = {"p": c["gnn_dropout"], "seed": drop["seed"], "stream": drop["stream_base"] + 32 + i}
            h = gat_conv(h, edge_index, p[f"gnn_layers.{i}.lin.weight"], p[f"gnn_layers.{i}.att_src"],
                         p[f"gnn_layers.{i}.att_dst"], p[f"gnn_layers.{i}.bias"], c["gnn_heads"], 0.2, adrop)
        elif t == "graphsage":
            h = sage_conv(h, edge_index, p[f"gnn_layers.{i}.lin_l.weight"], p[f"gnn_layers.{i}.lin_l.bias"],
                          p[f"gnn_layers.{i}.lin_r.weight"])
        elif t == "gin":
            h = gin_conv(h, edge_index, p[f"gnn_layers.{i}.eps"], p[f"gnn_layers.{i}.nn.0.weight"],
                         p[f"gnn_layers.{i}.nn.0.bias"], p[f"gnn_layers.{i}.nn.2.weight"], p[f"gnn_layers.{i}.nn.2.bias"])
        elif t in ("gatv2", "transformer", "fagcn"):
            adrop = None
            if drop is not None and c["gnn_dropout"] > 0:
                adrop = {"p": c["gnn_dropout"], "seed": drop["seed"], "stream": drop["stream_base"] + 32 + i}
            pre = f"gnn_layers.{i}"
            if t == "gatv2":
                h = gatv2_conv(h, edge_index, p[f"{pre}.lin_l.weight"], p[f"{pre}.lin_l.bias"], p[f"{pre}.lin_r.weight"],
                               p[f"{pre}.lin_r.bias"], p[f"{pre}.att"], p[f"{pre}.bias"], c["gnn_heads"], 0.2, adrop)
            elif t == "transformer":
                h = transformer_conv(h, edge_index, p, pre, c["gnn_heads"], adrop)
            else:
                h = fa_conv(h, x0, edge_index, p[f"{pre}.att_l.weight"], p[f"{pre}.att_r.weight"], 0.1, adrop)
        else:
            raise ValueError(f"Unsupported gnn_type: {t}")
        if c["use_layer_norm"]:
            h = F.layer_norm(h, (h.shape[1],), p[f"layer_norms.{i}.weight"], p[f"layer_norms.{i}.bias"])
        h = F.relu(h)
        if drop is not None:
            h = _mil.dropout(h, c["gnn_dropout"], drop["seed"], drop["stream_base"] + i,
                             elem_offset=int(drop.get("node_offset", 0)) * h.shape[1])
        if c["use_residual"] and h_prev.shape == h.shape:
            h = h + h_prev
        hs.append(h)
    atts, pooled = [], []
    for hd in range(c["att_heads"]):                   # 05:205-209
        tt = torch.tanh(F.linear(h, p[f"attention_layers.{hd}.0.weight"], p[f"attention_layers.{hd}.0.bias"]))
        a = torch.softmax(F.linear(tt, p[f"attention_layers.{hd}.2.weight"], p[f"attention_layers.{hd}.2.bias"]), dim=0)
        atts.append(a)
        pooled.append(torch.sum(a * h, dim=0))
    z = torch.stack(pooled, dim=0).mean(dim=0)         # 05:212
    att = torch.cat(atts, dim=1)                       # 05:213
    u = F.relu(F.linear(z, p["classifier.0.weight"], p["classifier.0.bias"]))   # light head, 05:133-139
    if drop is not None:
        u = _mil.dropout(u, c["pool_dropout"], drop["seed"], drop["stream_base"] + 64,
                         elem_offset=int(drop.get("graph_index", 0)) * u.shape[0])
    logits = F.linear(u, p["classifier.3.weight"], p["classifier.3.bias"])
    return {"probs": torch.softmax(logits, dim=0), "att": att, "hs": hs, "z": z, "logits": logits}


def graph_loss(probs, y):
    """`05_train_gnns.py:344`: CE(log(probs + 1e-9)[None], y).  Class weights of
    `05:328-331` cancel for a single sample and are omitted."""
    return F.cross_entropy(torch.log(probs + 1e-9).unsqueeze(0), torch.as_tensor(y).view(1).long())


def graphmil_loss_and_grads(p, cfg, x, edge_index, y, edge_weight=None):
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    xx = x.detach().clone().requires_grad_(True)
    out = graphmil_forward(q, cfg, xx, edge_index, edge_weight)
    loss = graph_loss(out["probs"], y)
    loss.backward()
    g = {k: (v.grad.detach() if v.grad is not None else torch.zeros_like(v)) for k, v in q.items()}
    g["x"] = xx.grad.detach()
    out = {k: ([t.detach() for t in v] if isinstance(v, list) else v.detach()) for k, v in out.items()}
    return loss.detach(), out, g
