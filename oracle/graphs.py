"""Patch-graph adjacency builders, restated on torch-CPU (oracle / test infrastructure).

Follows `03_build_graphs.py:15-78` and `utils_g_mil.py:564-674`.
"""
from __future__ import annotations

import math

import torch


def grid_edge_index(side=14, connect_diagonals=False):
    """`03_build_graphs.py:15-34`: lattice edges, node-major, offset order
    (-1,0),(1,0),(0,-1),(0,1)[,(-1,-1),(-1,1),(1,-1),(1,1)], no self loops."""
    offs = [(-1, 0), (1, 0), (0, -1), (0, 1)]
    if connect_diagonals:
        offs += [(-1, -1), (-1, 1), (1, -1), (1, 1)]
    src, dst = [], []
    for r in range(side):
        for c in range(side):
            for dr, dc in offs:
                rr, cc = r + dr, c + dc
                if 0 <= rr < side and 0 <= cc < side:
                    src.append(r * side + c)
                    dst.append(rr * side + cc)
    return torch.tensor([src, dst], dtype=torch.long)


def pairwise_sqdist(x):
    """`03_build_graphs.py:46-49`: ||x||^2 + ||x||^2^T - 2 x x^T, clamp >= 0, diag = +inf."""
    xn = (x ** 2).sum(dim=1, keepdim=True)
    d = xn + xn.t() - 2.0 * x @ x.t()
    d = torch.clamp(d, min=0.0)
    d.fill_diagonal_(float("inf"))
    return d


def knn_edge_index(x, k=8):
    """`03_build_graphs.py:37-54`: directed edges i -> nn(i), k clamped to [1, N-1],
    neighbours ascending by distance."""
    n = x.size(0)
    if n < 2:
        return torch.empty((2, 0), dtype=torch.long)
    k = int(max(1, min(k, n - 1)))
    _, nn_idx = torch.topk(pairwise_sqdist(x), k=k, dim=1, largest=False)
    src = torch.arange(n).unsqueeze(1).expand(-1, k)
    return torch.stack([src.reshape(-1), nn_idx.reshape(-1)], dim=0).long()


def random_edge_index(num_nodes, r=4, seed=None):
    """`03_build_graphs.py:57-78`: per node r targets from a seeded CPU
    ``torch.Generator`` permutation of the other nodes, symmetrised, then
    ``torch.unique(dim=1)`` (lexicographic)."""
    if num_nodes < 2:
        return torch.empty((2, 0), dtype=torch.long)
    r = int(max(1, min(r, num_nodes - 1)))
    g = torch.Generator()
    if seed is not None:
        g.manual_seed(int(seed))
    src, dst = [], []
    for i in range(num_nodes):
        perm = torch.randperm(num_nodes - 1, generator=g)[:r]
        chosen = perm + (perm >= i).long()  # i-th candidate list skips node i
        src += [i] * chosen.numel()
        dst += chosen.tolist()
    e = torch.tensor([src, dst], dtype=torch.long)
    e = torch.cat([e, e.flip(0)], dim=1)
    return torch.unique(e, dim=1)


def grid_adj(num_nodes, connect_diagonals=False):
    """`utils_g_mil.py:564-589`: dense A+I, row-normalised D^-1 (A+I); returns
    (adj_norm, adj_mask)."""
    s = int(math.isqrt(num_nodes))
    if s * s != num_nodes:
        raise ValueError("num_nodes must be a perfect square to build grid adjacency")
    e = grid_edge_index(s, connect_diagonals)
    adj = torch.zeros((num_nodes, num_nodes), dtype=torch.float32)
    adj[e[0], e[1]] = 1.0
    adj = adj + torch.eye(num_nodes)
    deg = adj.sum(dim=1)
    adj_norm = torch.diag(1.0 / deg) @ adj
    return adj_norm, (adj > 0).float()


def build_graph(x, graph_type="grid", k=None, connect_diagonals=False):
    """`utils_g_mil.py:618-674` (grid / knn branches): returns
    (adj_norm, adj_mask, edge_index, edge_weight)."""
    n = x.size(0)
    if graph_type == "grid":
        adj_norm, adj_mask = grid_adj(n, connect_diagonals)
        mask = adj_mask.bool()
        edge_index = mask.nonzero(as_tuple=False).t().long()
        return adj_norm, adj_mask, edge_index, adj_norm[mask]
    if graph_type == "knn":
        kk = 8 if k is None else int(k)
        # utils_g_mil.py:596-615: topk uses min(k, N-1)
        return None, None, knn_edge_index(x, min(kk, n - 1)), None
    raise ValueError(f"Unsupported graph_type='{graph_type}'. Supported types: 'grid', 'knn'.")
