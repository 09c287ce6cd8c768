"""``GraphMIL``: patch-graph GNN + multi-head attention pooling on the MI355X.

Same constructor, ``state_dict`` key names and ``forward(x, edge_index, edge_weight)
-> (probs, attention_weights)`` as the class the reference defines inside
`05_train_gnns.py:51-219`; message passing runs as a destination-major CSR SpMM
(``isic_spmm_csr_f32``), dense layers on exact-fp32 MFMA, LayerNorm/ReLU/dropout/
residual fused in one kernel, the 4-head attention pool in one launch per batch.

Graph models: ``mlp`` (pure dense, pinned by the reference), ``gcn``, ``gcnii``,
``graphsage`` (mean aggregation + L2 normalisation), ``gin`` (sum aggregation) and ``gat``
(per-destination edge softmax, the reference's tuned graph model) -- PyG ``GCNConv`` /
``GCN2Conv`` / ``SAGEConv`` / ``GINConv`` / ``GATConv`` semantics restated (``torch_geometric``
is absent and unpinned in the reference, see oracle/gnn.py), plus ``gatv2`` (GATv2Conv),
``transformer`` (TransformerConv(beta=True)) and ``fagcn`` (FAConv) on the edge-attention kernels of
``edge_attn.hip``.  Two of those cannot run as the reference writes them: its ``gatv2``
branch does not widen ``out_dim`` by the head count although ``concat=True`` (`05_train_gnns.py:99-101`),
so the following LayerNorm rejects the layer's output, and ``fagcn`` is called as
``layer(h, edge_index, edge_weight)`` (`:184-185`) whereas the layer takes ``(x, x_0, edge_index)``.
Here ``gatv2`` widens like ``gat`` and ``fagcn`` receives ``x_0`` = the projected input (as ``gcnii``).

Beyond the reference: ``forward`` also takes a batch of graphs (``offsets`` + global node ids,
or a prebuilt ``GraphBatch``) and returns ``probs[G, C]``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from isic_hip import ops
from isic_hip.bags import BagOffsets, as_offsets
from isic_hip.graph import (GraphBatch, fa_conv, gat_conv, gatv2_conv, gcn_block, l2_normalize, spmm,
                            transformer_attention)
from utils_g_mil import _DropoutClock

GNN_TYPES = ("mlp", "gcn", "gat", "gatv2", "gin", "graphsage", "transformer", "fagcn", "gcnii")
_BUILT = GNN_TYPES
# CSR the layer aggregates over: 'gcn' = self loops re-added (+ symmetric normalisation where the layer uses it),
# 'sum' / 'mean' = the edges as given (TransformerConv adds no self loops)
_GRAPH_MODE = {"gcn": "gcn", "gcnii": "gcn", "graphsage": "mean", "gin": "sum", "gat": "gcn", "gatv2": "gcn",
               "transformer": "sum", "fagcn": "gcn"}


class _GCNConvParams(nn.Module):
    """Parameter holder with PyG ``GCNConv`` names: ``bias`` [F], ``lin.weight`` [F, in] (Glorot / zeros)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(out_dim))
        self.lin = nn.Linear(in_dim, out_dim, bias=False)
        nn.init.xavier_uniform_(self.lin.weight)


class _GCN2ConvParams(nn.Module):
    """PyG ``GCN2Conv`` (shared weights): ``weight1`` [F, F], Glorot."""

    def __init__(self, channels, alpha, theta, layer):
        super().__init__()
        self.weight1 = nn.Parameter(torch.empty(channels, channels))
        nn.init.xavier_uniform_(self.weight1)
        self.alpha = float(alpha)
        self.beta = math.log(theta / layer + 1.0)


class _GATConvParams(nn.Module):
    """PyG ``GATConv(in, F, heads=H, concat)``: ``att_src`` / ``att_dst`` [1,H,F] (Glorot), ``bias``
    [H*F] (zeros; [F] when not concatenating), ``lin.weight`` [H*F, in] (Glorot, no bias)."""

    def __init__(self, in_dim, out_dim, heads, concat):
        super().__init__()
        self.att_src = nn.Parameter(torch.empty(1, heads, out_dim))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_dim))
        self.bias = nn.Parameter(torch.zeros(heads * out_dim if concat else out_dim))
        self.lin = nn.Linear(in_dim, heads * out_dim, bias=False)
        for t in (self.att_src, self.att_dst, self.lin.weight):
            nn.init.xavier_uniform_(t)
        self.heads, self.out_dim = heads, out_dim


class _GATv2ConvParams(nn.Module):
    """PyG ``GATv2Conv(in, F, heads=H, concat=True)``: ``att`` [1,H,F] (Glorot), ``bias`` [H*F] (zeros),
    ``lin_l`` / ``lin_r`` Linear(in, H*F) with bias (Glorot weights, zero biases)."""

    def __init__(self, in_dim, out_dim, heads):
        super().__init__()
        self.att = nn.Parameter(torch.empty(1, heads, out_dim))
        self.bias = nn.Parameter(torch.zeros(heads * out_dim))
        self.lin_l = nn.Linear(in_dim, heads * out_dim, bias=True)
        self.lin_r = nn.Linear(in_dim, heads * out_dim, bias=True)
        for t in (self.att, self.lin_l.weight, self.lin_r.weight):
            nn.init.xavier_uniform_(t)
        nn.init.zeros_(self.lin_l.bias)
        nn.init.zeros_(self.lin_r.bias)
        self.heads, self.out_dim = heads, out_dim


class _TransformerConvParams(nn.Module):
    """PyG ``TransformerConv(in, F, heads=H, concat=True, beta=True)``: ``lin_key`` / ``lin_query`` / ``lin_value`` /
    ``lin_skip`` Linear(in, H*F) with bias, ``lin_beta`` Linear(3*H*F, 1, bias=False)."""

    def __init__(self, in_dim, out_dim, heads):
        super().__init__()
        self.lin_key = nn.Linear(in_dim, heads * out_dim)
        self.lin_query = nn.Linear(in_dim, heads * out_dim)
        self.lin_value = nn.Linear(in_dim, heads * out_dim)
        self.lin_skip = nn.Linear(in_dim, heads * out_dim)
        self.lin_beta = nn.Linear(3 * heads * out_dim, 1, bias=False)
        self.heads, self.out_dim = heads, out_dim


class _FAConvParams(nn.Module):
    """PyG ``FAConv(channels, eps=0.1)``: ``att_l`` / ``att_r`` Linear(channels, 1, bias=False)."""

    def __init__(self, channels, eps=0.1):
        super().__init__()
        self.att_l = nn.Linear(channels, 1, bias=False)
        self.att_r = nn.Linear(channels, 1, bias=False)
        self.eps = float(eps)


class _SAGEConvParams(nn.Module):
    """PyG ``SAGEConv(aggr='mean', normalize=True)``: ``lin_l`` (neighbour mean, with bias) and
    ``lin_r`` (root, no bias)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.lin_l = nn.Linear(in_dim, out_dim, bias=True)
        self.lin_r = nn.Linear(in_dim, out_dim, bias=False)


class _GINConvParams(nn.Module):
    """PyG ``GINConv(nn, train_eps=True)``: ``eps`` [1] (init 0) and the wrapped ``nn`` Sequential."""

    def __init__(self, mlp):
        super().__init__()
        self.nn = mlp
        self.eps = nn.Parameter(torch.zeros(1))


class GraphMIL(nn.Module):
    def __init__(self, input_dim=768, gnn_type='gat', gnn_hidden=256, gnn_layers=2, gnn_dropout=0.1, gnn_heads=4,
                 gnn_concat=True, gcnii_alpha=0.1, gcnii_theta=0.5, att_dim=128, att_heads=4, pool_dropout=0.2,
                 classifier_dim=128, classifier_light=False, num_classes=7, use_residual=True, use_layer_norm=True):
        super().__init__()
        self.gnn_type = gnn_type.lower()
        if self.gnn_type not in GNN_TYPES:
            raise ValueError(f"Unsupported gnn_type: {self.gnn_type}")           # 05:113-114
        self.use_residual, self.use_layer_norm = use_residual, use_layer_norm
        self.classifier_light, self.gnn_heads, self.gnn_concat = classifier_light, gnn_heads, gnn_concat
        if (use_residual or self.gnn_type in {"fagcn", "gcnii"}) and input_dim != gnn_hidden:   # 05:65-68
            self.input_proj = nn.Linear(input_dim, gnn_hidden)
        else:
            self.input_proj = None
        self.gnn_layers = nn.ModuleList()
        self.layer_norms = nn.ModuleList() if use_layer_norm else None
        in_dim = input_dim if self.input_proj is None else gnn_hidden
        for i in range(gnn_layers):
            out_dim = gnn_hidden
            if self.gnn_type == 'gcn':
                layer = _GCNConvParams(in_dim, out_dim)
            elif self.gnn_type == 'gcnii':
                if in_dim != out_dim:
                    raise ValueError("GCNII requires a constant hidden dimension across layers")
                layer = _GCN2ConvParams(out_dim, gcnii_alpha, gcnii_theta, i + 1)
            elif self.gnn_type == 'gat':                                         # 05:83-86
                if not gnn_concat:
                    raise NotImplementedError("GATConv(concat=False) is not built on the HIP path")
                layer = _GATConvParams(in_dim, out_dim, gnn_heads, gnn_concat)
                out_dim *= gnn_heads
            elif self.gnn_type == 'gatv2':                                       # 05:99-101 (+ the widening of :86)
                if not gnn_concat:
                    raise NotImplementedError("GATv2Conv(concat=False) is not built on the HIP path")
                layer = _GATv2ConvParams(in_dim, out_dim, gnn_heads)
                out_dim *= gnn_heads
            elif self.gnn_type == 'transformer':                                 # 05:94-98
                if not gnn_concat:
                    raise NotImplementedError("TransformerConv(concat=False) is not built on the HIP path")
                layer = _TransformerConvParams(in_dim, out_dim, gnn_heads)
                out_dim *= gnn_heads
            elif self.gnn_type == 'fagcn':                                       # 05:102-105
                if in_dim != out_dim:
                    raise ValueError("FAGCN requires a constant hidden dimension")
                layer = _FAConvParams(out_dim, eps=0.1)
            elif self.gnn_type == 'graphsage':                                   # 05:87-88
                layer = _SAGEConvParams(in_dim, out_dim)
            elif self.gnn_type == 'gin':                                         # 05:89-93
                layer = _GINConvParams(nn.Sequential(nn.Linear(in_dim, out_dim), nn.ReLU(), nn.Linear(out_dim, out_dim)))
            else:
                layer = nn.Sequential(nn.Linear(in_dim, out_dim))
            self.gnn_layers.append(layer)
            if use_layer_norm:
                self.layer_norms.append(nn.LayerNorm(out_dim))
            in_dim = out_dim
        self.gnn_dropout = nn.Dropout(gnn_dropout)
        self.final_gnn_dim = in_dim
        self.att_heads = att_heads
        self.attention_layers = nn.ModuleList([
            nn.Sequential(nn.Linear(in_dim, att_dim), nn.Tanh(), nn.Linear(att_dim, 1)) for _ in range(att_heads)])
        if classifier_light:
            self.classifier = nn.Sequential(nn.Linear(in_dim, classifier_dim), nn.ReLU(), nn.Dropout(pool_dropout),
                                            nn.Linear(classifier_dim, num_classes))
        else:
            h2 = classifier_dim // 2
            self.classifier = nn.Sequential(
                nn.Linear(in_dim, classifier_dim), nn.LayerNorm(classifier_dim), nn.ReLU(), nn.Dropout(pool_dropout),
                nn.Linear(classifier_dim, h2), nn.LayerNorm(h2), nn.ReLU(), nn.Dropout(pool_dropout / 2),
                nn.Linear(h2, num_classes))
        self.dropout_clock = _DropoutClock()
        self.last_node_embeddings = None

    def set_dropout_state(self, seed, step=0):
        self.dropout_clock.seed, self.dropout_clock.step = int(seed), int(step)

    @property
    def graph_mode(self):
        """CSR aggregation mode the layers need (``GraphBatch`` mode), ``None`` for the graph-free 'mlp'."""
        return _GRAPH_MODE.get(self.gnn_type)

    def _gcnii_weight(self, i, layer):
        """GCN2Conv's effective weight ((1 - beta) I + beta W1)^T: the scaled identity is a constant of the layer, built once
        per device (no ``torch.eye`` per call); the one ``torch.add`` left carries the gradient into ``weight1``."""
        cache = self.__dict__.setdefault("_gcnii_eye", {})
        key = (i, layer.weight1.device)
        e = cache.get(key)
        if e is None:
            e = cache[key] = (1.0 - layer.beta) * torch.eye(layer.weight1.shape[0], device=layer.weight1.device,
                                                            dtype=torch.float32)
        return torch.add(e, layer.weight1, alpha=layer.beta).t()

    def _graph(self, edge_index, edge_weight, n_nodes, graph):
        if self.gnn_type == 'mlp':
            return None
        if graph is not None:
            if graph.mode != self.graph_mode:
                raise ValueError(f"gnn_type '{self.gnn_type}' needs a '{self.graph_mode}'-mode GraphBatch, "
                                 f"got '{graph.mode}'")
            return graph
        if edge_index is None:
            raise ValueError(f"gnn_type '{self.gnn_type}' needs edge_index")
        mode = _GRAPH_MODE[self.gnn_type]
        # 05:184-187 passes edge_weight to gcn / gcnii only
        return GraphBatch(edge_index, n_nodes, edge_weight if self.gnn_type in ("gcn", "gcnii") else None, mode=mode)

    def forward(self, x, edge_index=None, edge_weight=None, offsets=None, graph=None, labels=None, x_rows=None):
        """x[N, D] (+ edge_index[2, E]) -> (probs[C], attention_weights[N, heads]) as the reference
        (`05_train_gnns.py:156-219`); with ``offsets`` (graph boundaries, global node ids) the batch
        form returns probs[G, C].  With ``labels`` [G] the loss of the train loop (`05:344`,
        ``F.cross_entropy(log(probs + 1e-9), y)``) is the third return value -- and, for the classifier_light head in
        training mode, head + loss are one autograd node (``ops.graph_head_loss``: two launches instead of sixteen).
        ``x_rows = (rows, n_rows)``: ``x`` is a record store [R, D] and the batch's node row i is ``x[rows[i]]``
        (``train.GraphStore.batch_rows``): the input projection reads through the index, the gather never happens."""
        single = offsets is None
        if x_rows is not None and (offsets is None or self.input_proj is None):
            x = x[x_rows[0][:x_rows[1]].long()]            # no projection to read through the index: gather after all
            x_rows = None
        n_nodes = x_rows[1] if x_rows is not None else x.shape[0]
        offs = BagOffsets.single(n_nodes, x.device) if single else as_offsets(offsets, x.device)
        clk, tr = self.dropout_clock, self.training
        g = self._graph(edge_index, edge_weight, n_nodes, graph)
        if x_rows is not None:
            x_in = ops.linear_rows(x, x_rows[0], n_nodes, self.input_proj.weight, self.input_proj.bias)
        else:
            x_in = ops.linear(x, self.input_proj.weight, self.input_proj.bias) if self.input_proj is not None else x
        h, x0 = x_in, x_in
        p_drop = self.gnn_dropout.p
        for i, layer in enumerate(self.gnn_layers):
            h_prev = h
            if self.gnn_type == 'gcn' and self.use_layer_norm and self.use_residual and layer.lin.weight.shape[0] == h.shape[1]:
                # the reference's default layer (05:184-199) as one autograd node: the residual's and the convolution's
                # gradients into h meet in the data-gradient GEMM's epilogue
                ln = self.layer_norms[i]
                h = gcn_block(h, layer.lin.weight, layer.bias, ln.weight, ln.bias, g, ln.eps, clk.spec(p_drop, i, tr), True)
                continue
            if self.gnn_type == 'mlp':
                h = ops.linear(h, layer[0].weight, layer[0].bias)
            elif self.gnn_type == 'gcn':
                h = spmm(ops.linear(h, layer.lin.weight, None), g, bias=layer.bias)
            elif self.gnn_type == 'gat':           # edge softmax over the CSR rows; attention dropout = site 32 + i
                h = gat_conv(ops.linear(h, layer.lin.weight, None), layer.att_src, layer.att_dst, layer.bias, g,
                             layer.heads, 0.2, clk.spec(p_drop, 32 + i, tr))
            elif self.gnn_type == 'gatv2':         # edge softmax of att . leaky_relu(x_l[src] + x_r[dst]); dropout site 32 + i
                h = gatv2_conv(ops.linear(h, layer.lin_l.weight, layer.lin_l.bias),
                               ops.linear(h, layer.lin_r.weight, layer.lin_r.bias), layer.att, layer.bias, g, layer.heads,
                               0.2, clk.spec(p_drop, 32 + i, tr))
            elif self.gnn_type == 'transformer':   # scaled dot-product edge softmax + gated skip (beta)
                agg = transformer_attention(ops.linear(h, layer.lin_query.weight, layer.lin_query.bias),
                                            ops.linear(h, layer.lin_key.weight, layer.lin_key.bias),
                                            ops.linear(h, layer.lin_value.weight, layer.lin_value.bias), g, layer.heads,
                                            clk.spec(p_drop, 32 + i, tr))
                xr = ops.linear(h, layer.lin_skip.weight, layer.lin_skip.bias)
                # the scalar gate per node is index plumbing on [N, 3*H*F] -> [N, 1]: one fp32 GEMM + torch elementwise
                beta = torch.sigmoid(ops.linear(torch.cat([agg, xr, agg - xr], dim=1), layer.lin_beta.weight, None))
                h = beta * xr + (1.0 - beta) * agg
            elif self.gnn_type == 'fagcn':         # tanh-gated, GCN-normalised aggregation + eps * x0
                h = fa_conv(h, x0, layer.att_l.weight, layer.att_r.weight, g, layer.eps, clk.spec(p_drop, 32 + i, tr))
            elif self.gnn_type == 'graphsage':     # lin_l(mean_j x_j) + lin_r(x_i), then row L2 normalisation
                h = l2_normalize(ops.linear(spmm(h, g), layer.lin_l.weight, layer.lin_l.bias)
                                 + ops.linear(h, layer.lin_r.weight, None))
            elif self.gnn_type == 'gin':           # nn((1 + eps) x_i + sum_j x_j); the eps axpy is a torch op
                z0 = spmm(h, g) + (1.0 + layer.eps) * h
                z1 = ops.linear(z0, layer.nn[0].weight, layer.nn[0].bias, ops.ACT_RELU)
                h = ops.linear(z1, layer.nn[2].weight, layer.nn[2].bias)
            else:  # gcnii: (1-beta) p + beta p W1 with p = (1-alpha) A^ h + alpha x0  ==  p @ ((1-beta) I + beta W1)
                p = spmm(h, g, alpha=1.0 - layer.alpha, addend=x0, addend_scale=layer.alpha)
                h = ops.linear(p, self._gcnii_weight(i, layer), None)
            res = h_prev if (self.use_residual and h_prev.shape == h.shape) else None
            if self.use_layer_norm:
                ln = self.layer_norms[i]
                h = ops.layer_norm(h, ln.weight, ln.bias, ln.eps, relu=True, drop=clk.spec(p_drop, i, tr), residual=res)
            else:
                h = ops.relu_dropout(h, clk.spec(p_drop, i, tr))
                if res is not None:
                    h = h + res
        self.last_node_embeddings = h.detach()
        W2, b2, w3, b3 = ops.head_params(self.attention_layers)      # the heads' parameters as one [heads*A, H] operand
        z, att = ops.attn_pool(h, W2, b2, w3, b3, offs.device, offs.max_bag, heads=self.att_heads)
        c = self.classifier
        if (labels is not None and self.classifier_light and tr and not single
                and ops.graph_head_supported(z.shape[1], c[0].weight.shape[0], c[3].weight.shape[0])):
            probs, loss = ops.graph_head_loss(z, c[0].weight, c[0].bias, c[3].weight, c[3].bias, labels,
                                              clk.spec(c[2].p, 64, tr))
            clk.step += 1
            return probs, att, loss
        if self.classifier_light:
            u = ops.linear(z, c[0].weight, c[0].bias, ops.ACT_RELU, clk.spec(c[2].p, 64, tr))
            logits = ops.linear(u, c[3].weight, c[3].bias)
        else:
            u = ops.layer_norm(ops.linear(z, c[0].weight, c[0].bias), c[1].weight, c[1].bias, c[1].eps, relu=True,
                               drop=clk.spec(c[3].p, 64, tr))
            u = ops.layer_norm(ops.linear(u, c[4].weight, c[4].bias), c[5].weight, c[5].bias, c[5].eps, relu=True,
                               drop=clk.spec(c[7].p, 65, tr))
            logits = ops.linear(u, c[8].weight, c[8].bias)
        probs = ops.softmax_rows(logits)
        if tr:
            clk.step += 1
        if labels is not None:
            y = labels.reshape(-1)
            return (probs[0] if single else probs), att, ops.cross_entropy_from_probs(probs, y)
        return (probs[0], att) if single else (probs, att)
