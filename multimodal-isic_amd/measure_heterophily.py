"""Heterophily measures of the patch graphs on the MI355X -- the numeric core of the reference's
``04_measure_heterophily.py`` (`compute_edge_heterophily`, `:107-169`; `_summarize_image`, `:172-181`).

Per image and graph variant the reference computes, on CPU numpy: per-edge KL divergence of the teacher's patch
class distributions, per-edge Dirichlet energy of the patch embeddings, per-edge lattice distance, the adjusted
homophily of the teacher's dominant classes, the class compatibility matrix and the algebraic connectivity
lambda_2 of the symmetrised graph.  Here the edge-wise row gathers run in one HIP launch per batch of images
(``isic_edge_heterophily_f32``); the class bookkeeping is integer index plumbing in torch on the device, and
lambda_2 is a batched symmetric eigensolve on the device (``torch.linalg.eigvalsh``, a library call: the reference
uses ``np.linalg.eigvalsh``).  The reference's plotting / aggregation code (`:229-589`) is out of scope.
"""
from __future__ import annotations

import re

import numpy as np
import torch

from isic_hip.lib import IsicHipError, call

EPS = 1e-8                                                        # 04:11
MEASURES = ["H_kl", "H_dirichlet", "H_spatial", "H_adj", "lambda_2"]
GRAPH_VARIANT_RE = re.compile(r"^(?P<kind>grid4|grid8|knn|random)(?P<param>\d+)?$")
GRID_W = 14                                                       # 04:124-125 (% 14, // 14)


def edge_index_from_variant(row, graph_variant):
    """`_edge_index_from_variant` (04:87-104)."""
    m = GRAPH_VARIANT_RE.match(graph_variant)
    if not m:
        raise ValueError(f"Unsupported graph variant: {graph_variant}")
    kind, param = m.group("kind"), m.group("param")
    get = (lambda k: row[k]) if isinstance(row, dict) else (lambda k: getattr(row, k))
    if kind == "grid4":
        return np.asarray(get("grid4_edge_index"))
    if kind == "grid8":
        return np.asarray(get("grid8_edge_index"))
    if kind == "knn":
        return np.asarray(get("knn_edge_indices")[int(param)])
    return np.asarray(get("random_edge_indices")[int(param)])


def edge_measures(x, probs, dominant, edge_index, nodes_per_graph, grid_w=GRID_W, eps=EPS):
    """Device tensors of a batch of graphs -> (H_kl, H_dirichlet, H_spatial, same_class) per edge [E] (fp32), for
    ALL edges of ``edge_index[2, E]`` (global node ids; the caller drops self loops)."""
    for t in (x, probs, dominant, edge_index):
        if not t.is_cuda:
            raise IsicHipError("measure_heterophily runs on the MI355X only (no CPU fallback)")
    x = x.contiguous().float()
    probs = probs.contiguous().float()
    dom = dominant.contiguous().to(torch.int32)
    ei = edge_index.contiguous().to(torch.int64)
    E = int(ei.shape[1])
    out = torch.empty((4, E), device=x.device, dtype=torch.float32)
    call("isic_edge_heterophily_f32", x, probs, dom, ei[0], ei[1], E, int(x.shape[1]), int(probs.shape[1]), int(grid_w),
         int(nodes_per_graph), float(eps), out[0], out[1], out[2], out[3])
    return out[0], out[1], out[2], out[3]


def lambda2_batch(src, dst, n_graphs, nodes):
    """Second-smallest eigenvalue of I - D^-1/2 (A or A^T) D^-1/2 per graph (04:150-159); src/dst are global ids of
    the self-loop-free edges."""
    dev = src.device
    A = torch.zeros((n_graphs * nodes, nodes), device=dev, dtype=torch.float64)
    # duplicated edges count with their multiplicity: scipy's COO -> CSR conversion sums them (04:151-152)
    A.index_put_((src, dst % nodes), torch.ones(src.numel(), device=dev, dtype=torch.float64), accumulate=True)
    A = A.view(n_graphs, nodes, nodes)
    A = torch.maximum(A, A.transpose(1, 2))
    deg = A.sum(dim=2)
    dis = torch.where(deg > 0, deg.clamp_min(1e-300).rsqrt(), torch.zeros_like(deg))
    L = torch.eye(nodes, device=dev, dtype=torch.float64).unsqueeze(0) - dis.unsqueeze(2) * A * dis.unsqueeze(1)
    ev = torch.linalg.eigvalsh(L)
    return ev[:, 1] if nodes > 1 else torch.zeros(n_graphs, device=dev, dtype=torch.float64)


def compute_edge_heterophily_batch(embeddings, patch_probs, dominant_class, edge_indices, device="cuda:0"):
    """A batch of images of equal node count: lists of ``patch_embeddings[N,D]``, ``patch_probs[N,C]``,
    ``dominant_class[N]`` and per-image ``edge_index[2,E_i]`` (local ids).  Returns one dict per image with the
    reference's keys (`04:163-170`)."""
    dev = torch.device(device)
    n_img = len(embeddings)
    N = int(np.asarray(embeddings[0]).shape[0])
    C = int(np.asarray(patch_probs[0]).shape[1])
    x = torch.as_tensor(np.stack([np.asarray(e, dtype=np.float32) for e in embeddings])).to(dev).view(n_img * N, -1)
    p = torch.as_tensor(np.stack([np.asarray(q, dtype=np.float32) for q in patch_probs])).to(dev).view(n_img * N, C)
    dom = torch.as_tensor(np.stack([np.asarray(d, dtype=np.int32) for d in dominant_class])).to(dev).view(-1)
    eis = [torch.as_tensor(np.asarray(e, dtype=np.int64)) for e in edge_indices]
    counts = [int(e.shape[1]) for e in eis]
    ei = torch.cat([e + i * N for i, e in enumerate(eis)], dim=1).to(dev)
    gid = torch.repeat_interleave(torch.arange(n_img, device=dev), torch.as_tensor(counts, device=dev))
    kl, dirich, spatial, same = edge_measures(x, p, dom, ei, N)
    keep = ei[0] != ei[1]                                             # 04:117-118
    src, dst, gk = ei[0][keep], ei[1][keep], gid[keep]
    kl, dirich, spatial, same = kl[keep], dirich[keep], spatial[keep], same[keep]
    n_edges = torch.bincount(gk, minlength=n_img)
    edge_h = torch.zeros(n_img, device=dev, dtype=torch.float64).index_add_(0, gk, same.double()) / n_edges.clamp_min(1)
    pk = torch.zeros((n_img, C), device=dev, dtype=torch.float64)
    pk.index_put_((torch.arange(n_img * N, device=dev) // N, dom.long()), torch.ones(n_img * N, device=dev, dtype=torch.float64),
                  accumulate=True)
    pk /= max(1, N)
    expected = (pk * pk).sum(dim=1)
    h_adj = torch.where(expected < 1.0, (edge_h - expected) / (1.0 - expected).clamp_min(1e-300), torch.ones_like(expected))
    compat = torch.zeros((n_img, C, C), device=dev, dtype=torch.float64)
    compat.index_put_((gk, dom[src].long(), dom[dst].long()), torch.ones(src.numel(), device=dev, dtype=torch.float64),
                      accumulate=True)
    rs = compat.sum(dim=2, keepdim=True)
    compat = torch.where(rs != 0, compat / rs.clamp_min(1e-300), torch.zeros_like(compat))
    lam2 = lambda2_batch(src, dst, n_img, N)
    kl_c, di_c, sp_c = kl.cpu().numpy(), dirich.cpu().numpy(), spatial.cpu().numpy()
    bounds = np.concatenate([[0], np.cumsum(n_edges.cpu().numpy())])
    edge_h_c, h_adj_c, compat_c, lam_c = edge_h.cpu().numpy(), h_adj.cpu().numpy(), compat.cpu().numpy(), lam2.cpu().numpy()
    out = []
    for i in range(n_img):
        a, b = int(bounds[i]), int(bounds[i + 1])
        out.append({"H_kl": kl_c[a:b], "H_dirichlet": di_c[a:b], "H_spatial": sp_c[a:b],
                    "H_adj": float(h_adj_c[i]), "lambda_2": np.array([float(lam_c[i])]),
                    "H_compat_matrix": compat_c[i]})
        if b == a:                                                    # 04:131: mean of an empty edge set -> 0.0
            e0 = float(expected[i])
            out[-1]["H_adj"] = (0.0 - e0) / (1.0 - e0) if e0 < 1.0 else 1.0
    return out


def compute_edge_heterophily(row, graph_variant=None, device="cuda:0"):
    """Drop-in for the reference function (04:107): one image (a row with ``patch_embeddings``, ``patch_probs``,
    ``dominant_class`` and either ``edge_index`` or the 03 graph columns)."""
    get = (lambda k: row[k]) if isinstance(row, dict) else (lambda k: getattr(row, k))
    ei = np.asarray(get("edge_index")) if graph_variant is None else edge_index_from_variant(row, graph_variant)
    return compute_edge_heterophily_batch([get("patch_embeddings")], [get("patch_probs")], [get("dominant_class")], [ei],
                                          device=device)[0]


def summarize_image(em, meta):
    """`_summarize_image` (04:172-181)."""
    out = dict(meta)
    out["num_edges"] = len(em["H_kl"])
    for m in MEASURES:
        vals = em[m]
        out[f"{m}_mean"] = float(np.mean(vals))
        out[f"{m}_std"] = float(np.std(vals))
        out[f"{m}_median"] = float(np.median(vals))
    out["H_compat_matrix"] = em["H_compat_matrix"]
    return out
