"""MI355X-native drop-in for the adjacency builders of the reference's
``03_build_graphs.py`` (`_grid_edge_index`, `_knn_edge_index`, `_random_edge_index`,
`_build_image_graph_blueprints`, `process_model_directory`) with the same return
types, edge ordering and pickle schema.

The k-NN build -- the hot part, run 10x per image by the reference
(`03_build_graphs.py:104-105`) -- is one HIP launch for a whole batch of images:
exact-fp32 MFMA distance tiles + per-row top-k in LDS; neighbour lists for every
requested k come from ONE top-16 (a k-NN list is a prefix of a k'-NN list).
"""
from __future__ import annotations

import pickle
from pathlib import Path

import numpy as np
import pandas as pd
import torch

from isic_hip.bags import BagOffsets
from isic_hip.graph import knn_indices

NUM_NODES = 196
GRID_SIDE = 14
DEFAULT_K_VALUES = tuple(range(1, 9)) + (12, 16)
DEFAULT_R_VALUES = tuple(range(1, 9)) + (12, 16)
_GRID_CACHE = {}


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("build_graphs needs the MI355X: the k-NN build has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _grid_edge_index(connect_diagonals=False, side=GRID_SIDE):
    """Lattice edges, node-major, neighbour order up/down/left/right[/diagonals], no self
    loops (`03_build_graphs.py:15-34`).  Integer-only host work, cached."""
    key = (bool(connect_diagonals), int(side))
    if key not in _GRID_CACHE:
        if side * side != NUM_NODES and side == GRID_SIDE:
            raise ValueError("NUM_NODES must be a perfect square for grid graphs")
        steps = [(-1, 0), (1, 0), (0, -1), (0, 1)] + ([(-1, -1), (-1, 1), (1, -1), (1, 1)] if connect_diagonals else [])
        r, c = np.divmod(np.arange(side * side), side)
        src, dst = [], []
        for dr, dc in steps:
            rr, cc = r + dr, c + dc
            ok = (rr >= 0) & (rr < side) & (cc >= 0) & (cc < side)
            src.append(np.where(ok, r * side + c, -1))
            dst.append(np.where(ok, rr * side + cc, -1))
        src, dst = np.stack(src, 1).reshape(-1), np.stack(dst, 1).reshape(-1)   # node-major, step-minor
        keep = src >= 0
        _GRID_CACHE[key] = torch.from_numpy(np.stack([src[keep], dst[keep]]).astype(np.int64))
    return _GRID_CACHE[key].clone()


def knn_edge_index_batched(x, offsets, k_values):
    """x[sum N, D] (device or host) + ragged graph offsets -> {k: edge_index[2, sum N*k_g]} with
    LOCAL node ids per graph when a single graph is given, GLOBAL ids otherwise."""
    dev = x.device if x.is_cuda else _device()
    offs = offsets if isinstance(offsets, BagOffsets) else BagOffsets(offsets, dev)
    n_min = int(np.diff(offs.host).min()) if offs.num_bags else 0
    out = {}
    if offs.total == 0 or n_min < 2:
        if offs.num_bags <= 1:
            return {int(k): torch.empty((2, 0), dtype=torch.long) for k in k_values}
        raise ValueError("every graph of a batch needs at least 2 nodes")
    ks = {int(k): int(max(1, min(int(k), n_min - 1))) for k in k_values}     # 03:45 clamp
    kmax = max(ks.values())
    nn = knn_indices(x.to(dev, torch.float32), offs, kmax)                   # [T, kmax] local ids
    base = torch.from_numpy(np.repeat(offs.host[:-1], np.diff(offs.host))).to(dev)
    src_all = torch.arange(offs.total, device=dev)
    for k, kk in ks.items():
        dst = nn[:, :kk] + (base[:, None] if offs.num_bags > 1 else 0)
        src = (src_all if offs.num_bags > 1 else src_all)[:, None].expand(-1, kk)
        out[k] = torch.stack([src.reshape(-1), dst.reshape(-1)], dim=0).long()
    return out


def _knn_edge_index(x, k=8):
    """Directed k-NN edges i -> nn(i), neighbours ascending by distance
    (`03_build_graphs.py:37-54`).  Returned on ``x``'s own device, like the reference."""
    if x.ndim != 2:
        raise ValueError("x must be a 2D tensor [num_nodes, feature_dim]")
    n = x.size(0)
    if n < 2:
        return torch.empty((2, 0), dtype=torch.long)
    e = knn_edge_index_batched(x, [0, n], [k])[int(k)]
    return e if x.is_cuda else e.cpu()


def _random_edge_index(num_nodes, r=4, seed=None):
    """Random undirected graph of `03_build_graphs.py:57-78`: per node ``r`` targets from a seeded
    CPU ``torch.Generator`` permutation, symmetrised, deduplicated in lexicographic order.  Stays on
    the host: it is defined by torch's CPU random stream."""
    if num_nodes < 2:
        return torch.empty((2, 0), dtype=torch.long)
    r = int(max(1, min(r, num_nodes - 1)))
    gen = torch.Generator()
    if seed is not None:
        gen.manual_seed(int(seed))
    picks = torch.stack([torch.randperm(num_nodes - 1, generator=gen)[:r] for _ in range(num_nodes)])   # [N, r]
    owner = torch.arange(num_nodes).unsqueeze(1)
    picks = picks + (picks >= owner).long()          # candidate list of node i skips i itself
    e = torch.stack([owner.expand(-1, r).reshape(-1), picks.reshape(-1)])
    return torch.unique(torch.cat([e, e.flip(0)], dim=1), dim=1)


def _as_list(value):
    return list(value) if isinstance(value, (list, tuple, set)) else [value]


def _parse_patch_stats_filename(path):
    parts = Path(path).stem.split("_")
    if len(parts) < 5 or parts[:3] != ["patch", "stats", "fold"]:
        raise ValueError(f"Unexpected patch-stats filename: {path}")
    return int(parts[3]), parts[4]


def _build_image_graph_blueprints(row, k_values, r_values, seed):
    """One image -> dict of numpy edge arrays (`03_build_graphs.py:95-114`)."""
    x = torch.as_tensor(row["patch_embeddings"], dtype=torch.float32)
    knn = knn_edge_index_batched(x, [0, x.shape[0]], [int(k) for k in k_values])
    return {
        "grid4_edge_index": _grid_edge_index(False).numpy(),
        "grid8_edge_index": _grid_edge_index(True).numpy(),
        "knn_edge_indices": {int(k): knn[int(k)].cpu().numpy() for k in k_values},
        "random_edge_indices": {int(r): _random_edge_index(NUM_NODES, r=int(r), seed=seed).numpy() for r in r_values},
    }


def build_graph_records(teacher_df, model_name, fold, split, k_values, r_values, seed):
    """All images of one patch-stats frame in ONE k-NN launch (same records as the per-row loop)."""
    xs = [np.asarray(v, dtype=np.float32) for v in teacher_df["patch_embeddings"]]
    offs = np.concatenate([[0], np.cumsum([a.shape[0] for a in xs])])
    knn = knn_edge_index_batched(torch.from_numpy(np.concatenate(xs)), offs, [int(k) for k in k_values])
    knn = {k: v.cpu().numpy() for k, v in knn.items()}
    records = []
    for i, (row_idx, row) in enumerate(teacher_df.iterrows()):
        lo, n = int(offs[i]), int(offs[i + 1] - offs[i])
        per_k = {}
        for k in k_values:
            kk = knn[int(k)].shape[1] // int(offs[-1])
            per_k[int(k)] = knn[int(k)][:, lo * kk:(lo + n) * kk] - (lo if len(xs) > 1 else 0)
        records.append({
            "model_name": model_name, "fold": fold, "split": split, "image_id": row["image_id"],
            "grid4_edge_index": _grid_edge_index(False).numpy(),
            "grid8_edge_index": _grid_edge_index(True).numpy(),
            "knn_edge_indices": per_k,
            "random_edge_indices": {int(r): _random_edge_index(NUM_NODES, r=int(r), seed=seed + fold * 10_000 + row_idx).numpy()
                                    for r in r_values},
        })
    return records


def process_model_directory(model_dir, output_root, k_values=DEFAULT_K_VALUES, r_values=DEFAULT_R_VALUES, seed=42):
    """`03_build_graphs.py:117-149`: patch_stats_fold_*_*.pkl -> graph_outputs/<model>/graph_dataset.pkl."""
    model_dir, output_root = Path(model_dir), Path(output_root)
    out_dir = output_root / model_dir.name
    out_dir.mkdir(parents=True, exist_ok=True)
    records = []
    for path in sorted(model_dir.glob("patch_stats_fold_*_*.pkl")):
        fold, split = _parse_patch_stats_filename(path)
        with open(path, "rb") as f:
            frame = pickle.load(f)
        if not isinstance(frame, pd.DataFrame):
            frame = pd.DataFrame(frame)
        records += build_graph_records(frame, model_dir.name, fold, split, _as_list(k_values), _as_list(r_values), seed)
    out_path = out_dir / "graph_dataset.pkl"
    with open(out_path, "wb") as f:
        pickle.dump(pd.DataFrame(records), f)
    print(f"Saved: {out_path}")
    return out_path
