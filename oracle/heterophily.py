"""Heterophily measures of one patch graph, restated in numpy (oracle / test infrastructure).

Follows ``compute_edge_heterophily`` of `04_measure_heterophily.py:107-169` measure by measure; pinned by
``tests/golden/heterophily.npz``, which oracle/gen_golden.py produces with the reference's own function.
lambda_2 uses a dense symmetric eigensolve exactly as the reference (`np.linalg.eigvalsh`, 04:158)."""
from __future__ import annotations

import numpy as np

EPS = 1e-8            # 04:11
GRID_W = 14           # 04:124-125


def edge_heterophily(embeddings, patch_probs, dominant_class, edge_index):
    src, dst = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    keep = src != dst                                                          # 04:117-118
    src, dst = src[keep], dst[keep]
    emb = np.asarray(embeddings, dtype=np.float32)
    p = np.asarray(patch_probs, dtype=np.float32)
    dom = np.asarray(dominant_class, dtype=np.int32)
    n, c = emb.shape[0], p.shape[1]
    same = dom[src] == dom[dst]
    edge_h = float(np.mean(same)) if len(same) > 0 else 0.0                     # 04:130-131
    pk = np.bincount(dom, minlength=c) / max(1, n)
    expected = float(np.sum(pk ** 2))                                          # 04:132-134
    h_adj = (edge_h - expected) / (1.0 - expected) if expected < 1.0 else 1.0   # 04:136-139
    compat = np.zeros((c, c), dtype=np.float64)
    np.add.at(compat, (dom[src], dom[dst]), 1.0)                               # 04:141-145
    rs = compat.sum(axis=1, keepdims=True)
    compat = np.divide(compat, rs, out=np.zeros_like(compat), where=rs != 0)    # 04:146-152
    A = np.zeros((n, n), dtype=np.float64)
    np.add.at(A, (src, dst), 1.0)                # 04:151: scipy's COO -> CSR conversion SUMS duplicated edges
    A = np.maximum(A, A.T)                                                     # 04:152
    deg = A.sum(axis=1)
    dis = np.zeros_like(deg)
    dis[deg > 0] = 1.0 / np.sqrt(deg[deg > 0])
    L = np.eye(n) - dis[:, None] * A * dis[None, :]                            # 04:157-160
    ev = np.linalg.eigvalsh(L)
    lam2 = float(np.sort(ev)[1]) if len(ev) > 1 else 0.0
    xs, ys = src % GRID_W, src // GRID_W
    xd, yd = dst % GRID_W, dst // GRID_W
    return {
        "H_kl": np.sum(p[src] * np.log((p[src] + EPS) / (p[dst] + EPS)), axis=1),           # 04:164
        "H_dirichlet": 0.5 * np.sum((emb[src] - emb[dst]) ** 2, axis=1),                     # 04:165
        "H_spatial": np.sqrt((xs - xd) ** 2 + (ys - yd) ** 2),                               # 04:166
        "H_adj": h_adj, "lambda_2": np.array([lam2]), "H_compat_matrix": compat,
    }
