"""MI355X-native drop-in for the MIL / graph-MIL building blocks of the reference's
``utils_g_mil.py`` (same import surface: ``from utils_g_mil import
AttentionMIL_teacher, AttentionMIL, PatientDataset, GraphMIL, build_graph``).

The classes keep the reference constructors, ``state_dict`` key names and return
values (reference `utils_g_mil.py:15-114`), but ``forward`` runs hand-written HIP
kernels through ``libisic_hip.so`` and additionally accepts a whole batch of
ragged bags (``x[sum K, D]`` + ``offsets[B+1]``) in one launch.  CPU tensors are
rejected: there is no PyTorch fallback.
"""
from __future__ import annotations

import random

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import Dataset

from isic_hip import ops
from isic_hip.bags import BagOffsets, as_offsets


class _DropoutClock:
    """Counter-based dropout state shared by a module's dropout sites: stream id =
    step * 1024 + site (+ rank << 44 under data parallelism), so every (rank, step, site, element)
    draws an independent word.  Element indices are local to a rank's shard of the step's bags, so
    without the rank term every shard would see the same mask; with it the shards' masks are
    independent, as the rows of one big batch are."""

    def __init__(self, seed=None, rank=None):
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self.step = 0
        self.rank = rank       # None: torch.distributed rank at use time (0 when not initialised)
        # device step clock (isic_hip.graphs.StepClock.tensor): the step then lives in HBM and the kernels add it to the
        # stream id themselves -- what lets a whole train step be captured into a hipGraph and replayed; ``step`` is
        # ignored while it is set
        self.device_clock = None

    def _rank(self):
        if self.rank is not None:
            return int(self.rank)
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    def spec(self, p, site, training):
        if not training or p <= 0.0:
            return None
        if self.device_clock is not None:
            return ops.DropoutSpec(p, self.seed, (self._rank() << 44) + site, clock=self.device_clock)
        return ops.DropoutSpec(p, self.seed, (self._rank() << 44) + self.step * 1024 + site)


class _MILBase(nn.Module):
    def __init__(self, input_dim, hidden_dim, att_dim, dropout):
        super().__init__()
        # parameter containers with the reference's names / default init (utils_g_mil.py:18-27, 49-59)
        self.feature_extractor = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.ReLU(), nn.Dropout(dropout))
        self.attention = nn.Sequential(nn.Linear(hidden_dim, att_dim), nn.Tanh(), nn.Linear(att_dim, 1))
        self.dropout_p = float(dropout)
        self.dropout_clock = _DropoutClock()

    def set_dropout_state(self, seed, step=0):
        self.dropout_clock.seed, self.dropout_clock.step = int(seed), int(step)

    def _hidden(self, x):
        fe = self.feature_extractor[0]
        drop = self.dropout_clock.spec(self.dropout_p, 0, self.training)
        h = ops.linear(x, fe.weight, fe.bias, ops.ACT_RELU, drop)
        if self.training:
            self.dropout_clock.step += 1
        return h


class AttentionMIL_teacher(_MILBase):
    """Reference `utils_g_mil.py:38-105`: class-space attention pooling.

    ``forward(x)`` with ``x[N, D]`` returns the reference's 5-key dict for one bag;
    ``forward(x, offsets)`` with ``x[sum K, D]`` returns the same keys batched
    (``bag_logits[B, C]``, ``attention[sum K]`` ...).
    """

    def __init__(self, input_dim=768, hidden_dim=128, att_dim=64, dropout=0.5, num_classes=7):
        super().__init__(input_dim, hidden_dim, att_dim, dropout)
        self.patch_classifier = nn.Linear(hidden_dim, num_classes)

    def forward(self, x, offsets=None, return_pooled=False):
        single = offsets is None
        offs = BagOffsets.single(x.shape[0], x.device) if single else as_offsets(offsets, x.device)
        h = self._hidden(x)
        a0, a2 = self.attention[0], self.attention[2]
        z, att, P, PP, BL, BP = ops.attn_pool(h, a0.weight, a0.bias, a2.weight, a2.bias, offs.device, offs.max_bag,
                                              heads=1, W4=self.patch_classifier.weight,
                                              b4=self.patch_classifier.bias)
        out = {
            "bag_logits": BL[0] if single else BL,
            "bag_probs": BP[0] if single else BP,
            "attention": att.reshape(-1),
            "patch_logits": P,
            "patch_probs": PP,
        }
        if return_pooled:
            out["pooled"] = z[0] if single else z
            out["hidden"] = h
        return out


class AttentionMIL(_MILBase):
    """Reference `utils_g_mil.py:15-36`: feature-space pooling; returns ``(probs, a)``."""

    def __init__(self, input_dim=76, hidden_dim=128, att_dim=64, dropout=0.5, num_classes=7):
        super().__init__(input_dim, hidden_dim, att_dim, dropout)
        self.classifier = nn.Linear(hidden_dim, num_classes)

    def forward(self, x, offsets=None):
        single = offsets is None
        offs = BagOffsets.single(x.shape[0], x.device) if single else as_offsets(offsets, x.device)
        h = self._hidden(x)
        a0, a2 = self.attention[0], self.attention[2]
        z, att = ops.attn_pool(h, a0.weight, a0.bias, a2.weight, a2.bias, offs.device, offs.max_bag, heads=1)
        logits = ops.linear(z, self.classifier.weight, self.classifier.bias)
        probs = ops.softmax_rows(logits)
        if single:
            return probs[0], att            # probs[C], a[N,1]
        return probs, att


class PatientDataset(Dataset):
    """Reference `utils_g_mil.py:107-114` (same item contract: features, float label)."""

    def __init__(self, patient_features, patient_labels):
        self.features = patient_features
        self.labels = patient_labels

    def __len__(self):
        return len(self.features)

    def __getitem__(self, idx):
        return self.features[idx], torch.tensor(self.labels[idx], dtype=torch.float32)


def set_seed(seed: int = 42):
    """Reference `utils_g_mil.py:116-123`."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


# ----------------------------------------------------------------------------- graph builders / graph-MIL
_GRID_ADJ_CACHE = {}


def build_grid_adj(num_nodes, connect_diagonals=False, device=None):
    """Reference `utils_g_mil.py:564-589`: dense (A + I) row-normalised by degree; returns
    ``(adj_norm, adj_mask)``.  Constant integer geometry: built once on the host."""
    s = int(np.sqrt(num_nodes))
    if s * s != num_nodes:
        raise ValueError('num_nodes must be a perfect square to build grid adjacency')
    from build_graphs import _grid_edge_index
    e = _grid_edge_index(connect_diagonals, side=s)
    adj = torch.zeros((num_nodes, num_nodes), dtype=torch.float32)
    adj[e[0], e[1]] = 1.0
    adj = adj + torch.eye(num_nodes)
    adj_norm = adj / adj.sum(dim=1, keepdim=True)
    if device is not None:
        adj_norm = adj_norm.to(device)
    return adj_norm, (adj > 0).float()


def build_knn_edge_index(x, k=8, device=None):
    """Reference `utils_g_mil.py:596-615`: k-NN ``edge_index[2, N*k]`` (HIP distance + top-k)."""
    from build_graphs import knn_edge_index_batched
    n = x.size(0)
    e = knn_edge_index_batched(x, [0, n], [min(int(k), n - 1)])[min(int(k), n - 1)]
    return e.to(x.device if device is None else device)


def build_graph(x, graph_type='grid', k=None, connect_diagonals=False, device=None):
    """Reference `utils_g_mil.py:618-674`: ``(adj_norm, adj_mask, edge_index, edge_weight)``."""
    num_nodes = x.size(0)
    if graph_type == 'grid':
        key = (num_nodes, bool(connect_diagonals), str(device))
        if key not in _GRID_ADJ_CACHE:
            _GRID_ADJ_CACHE[key] = build_grid_adj(num_nodes, connect_diagonals=connect_diagonals, device=device)
        adj_norm, adj_mask = _GRID_ADJ_CACHE[key]
        mask = adj_mask.bool()
        edge_index = mask.nonzero(as_tuple=False).t().to(x.device).long()
        edge_weight = adj_norm.to(mask.device)[mask].to(x.device)
        return adj_norm, adj_mask, edge_index, edge_weight
    if graph_type == 'knn':
        return None, None, build_knn_edge_index(x, k=8 if k is None else int(k), device=device), None
    if graph_type == 'random':
        from build_graphs import _random_edge_index
        return None, None, _random_edge_index(num_nodes, r=k if k is not None else 4).to(x.device), None
    raise ValueError(f"Unsupported graph_type='{graph_type}'. Supported types: 'grid', 'knn'.")


def __getattr__(name):
    # ``from utils_g_mil import GraphMIL`` (use_latent.py:21): the older copy of the class takes
    # (x, adj, adj_mask, edge_index, edge_weight); resolved lazily to avoid a circular import.
    if name == "GraphMIL":
        from gnn_models import GraphMIL as _G

        class GraphMIL(_G):
            def forward(self, x, adj=None, adj_mask=None, edge_index=None, edge_weight=None):
                return super().forward(x, edge_index=edge_index, edge_weight=edge_weight)

        globals()["GraphMIL"] = GraphMIL
        return GraphMIL
    raise AttributeError(name)
