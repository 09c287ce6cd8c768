"""Teacher -> patch statistics -> graphs -> GNN WITHOUT leaving HBM.

The reference hands the stages over through pickled DataFrames on disk: 01 writes
``teacher_outputs_fold_{f}_{split}.pkl`` (`01_train_mil_teacher.py:69-87,303-305`), 02 adds
``dominant_class = argmax(patch_probs)`` (`02_compute_patch_statistics.py:17`), 03 rebuilds the distance
matrix of every image ten times for its ten k values (`03_build_graphs.py:104-105`) and 05 re-uploads ``x`` and
``edge_index`` of every graph at every step (`05_train_gnns.py:340-343`).  Here the teacher's outputs stay on the
device: one batched eval forward, one arg-max, ONE k-NN launch for all images whose top-16 serves every k (the
k-NN of a smaller k is a prefix of the larger one: neighbours are sorted by distance), and the GNN loop reads its
graphs straight from those tensors (``graph_records`` -> ``train.GraphStore``).  The three pickle schemas are
kept as EXPORTS (``teacher_frame`` / ``patch_stats_frame`` / ``graph_frame``), byte-compatible with the drop-in
scripts ``01`` / ``02`` / ``03``.
"""
from __future__ import annotations

import numpy as np
import torch

import build_graphs as bg
from isic_hip.bags import BagOffsets
from isic_hip.graph import knn_indices
from isic_hip.lib import IsicHipError


class DeviceTeacherOutputs:
    """Teacher outputs of G images of N patches, resident in HBM:
    ``x[G,N,D]``, ``patch_probs[G,N,C]``, ``attention[G,N]``, ``dominant_class[G,N]`` (int32), ``labels[G]``
    (int64), ``knn[G,N,kmax]`` (local neighbour ids, ascending by distance)."""

    def __init__(self, x, patch_probs, attention, labels, image_ids, kmax=16):
        if not x.is_cuda:
            raise IsicHipError("the on-device pipeline needs device tensors (no CPU fallback)")
        G, N, D = x.shape
        self.x, self.patch_probs, self.attention = x, patch_probs, attention
        self.labels = labels
        self.image_ids = list(image_ids)
        self.dominant_class = patch_probs.argmax(dim=2).to(torch.int32)                       # 02:17
        self.kmax = int(max(1, min(kmax, N - 1)))
        offs = BagOffsets.uniform(G, N, x.device)
        self.knn = knn_indices(x.reshape(G * N, D), offs, self.kmax).view(G, N, self.kmax)    # 03:46-50, all k at once

    def __len__(self):
        return int(self.x.shape[0])

    def knn_edge_index(self, k):
        """[G, 2, N*k] int64, local node ids, edge order of `03_build_graphs.py:52-53` (source-major)."""
        G, N, _ = self.x.shape
        k = int(max(1, min(int(k), N - 1)))                                                    # 03:45
        if k > self.kmax:
            raise ValueError(f"k = {k} exceeds the resident top-{self.kmax}")
        src = torch.arange(N, device=self.x.device).view(1, N, 1).expand(G, N, k)
        return torch.stack([src.reshape(G, -1), self.knn[:, :, :k].reshape(G, -1)], dim=1)

    def edge_index(self, variant, fold=0, seed=42):
        """Edges of a graph variant name of 05 (`05_train_gnns.py:228-239`): knn<k> on the device; grid4 / grid8 are
        one constant lattice; random<r> is defined by torch's CPU random stream (`03:57-78`) and built on the host."""
        G, N, _ = self.x.shape
        dev = self.x.device
        if variant.startswith("knn"):
            return self.knn_edge_index(int(variant[3:]))
        if variant in ("grid4", "grid8"):
            e = bg._grid_edge_index(variant == "grid8").to(dev)
            return e.unsqueeze(0).expand(G, -1, -1)
        if variant.startswith("random"):
            r = int(variant[6:])
            es = [bg._random_edge_index(N, r=r, seed=seed + fold * 10_000 + i) for i in range(G)]
            if len({int(e.shape[1]) for e in es}) != 1:
                return [e.to(dev) for e in es]                 # ragged: a list, one tensor per image
            return torch.stack(es).to(dev)
        raise ValueError(f"Unsupported graph variant: {variant}")

    def graph_records(self, variant, fold=0, seed=42):
        """Records for ``train.GraphStore`` / ``train_gnn_fold`` (`05_train_gnns.py:248-270` schema: x, edge_index, y)
        -- views of the resident tensors, nothing is copied to the host."""
        ei = self.edge_index(variant, fold, seed)
        y = self.labels.tolist()
        return [{"x": self.x[i], "edge_index": ei[i], "y": int(y[i]), "image_id": self.image_ids[i]} for i in range(len(self))]

    # ------------------------------------------------------------------ exports (the reference's pickle schemas)
    def teacher_frame(self):
        """`01_train_mil_teacher.py:69-87`: image_id, label, patch_probs, attention, patch_embeddings."""
        import pandas as pd
        pp, att, x, y = (t.cpu().numpy() for t in (self.patch_probs, self.attention, self.x, self.labels))
        return pd.DataFrame([{"image_id": self.image_ids[i], "label": int(y[i]), "patch_probs": pp[i], "attention": att[i],
                              "patch_embeddings": x[i]} for i in range(len(self))])

    def patch_stats_frame(self):
        """`02_compute_patch_statistics.py:19-26`: image_id, label, patch_embeddings, patch_probs, dominant_class."""
        import pandas as pd
        pp, x, y = (t.cpu().numpy() for t in (self.patch_probs, self.x, self.labels))
        dom = self.dominant_class.cpu().numpy().astype(np.int64)
        return pd.DataFrame({"image_id": self.image_ids, "label": [int(v) for v in y], "patch_embeddings": list(x),
                             "patch_probs": list(pp), "dominant_class": list(dom)})

    def graph_frame(self, model_name, fold, split, k_values=bg.DEFAULT_K_VALUES, r_values=bg.DEFAULT_R_VALUES, seed=42,
                    row_offset=0):
        """`03_build_graphs.py:95-149`: one row per image with grid / k-NN / random edge arrays (numpy)."""
        import pandas as pd
        N = int(self.x.shape[1])
        knn = {int(k): self.knn_edge_index(k).cpu().numpy() for k in k_values}
        g4, g8 = bg._grid_edge_index(False).numpy(), bg._grid_edge_index(True).numpy()
        rows = []
        for i in range(len(self)):
            rows.append({"model_name": model_name, "fold": fold, "split": split, "image_id": self.image_ids[i],
                         "grid4_edge_index": g4, "grid8_edge_index": g8,
                         "knn_edge_indices": {k: v[i] for k, v in knn.items()},
                         "random_edge_indices": {int(r): bg._random_edge_index(N, r=int(r), seed=seed + fold * 10_000 + row_offset + i).numpy()
                                                 for r in r_values}})
        return pd.DataFrame(rows)


@torch.no_grad()
def with_radiomic_node_features(x, radiomics):
    """BASELINE.json configs[3]: "k-NN graphs on patch embeddings + radiomic node feats".  The reference's graph records
    carry the patch embeddings only (`05_train_gnns.py:248-270`: ``x[196, 768]``) and its radiomics are zero-stubbed
    (`dataset.py:42`), so there is no reference layout to follow: the lesion-level radiomic vector is appended to EVERY
    node of its graph (``x[G, N, D]``, ``radiomics[G, R]`` -> ``[G, N, D + R]``); the k-NN edges stay those of the patch
    embeddings.  Stays on the device."""
    if x.dim() != 3 or radiomics.dim() != 2 or radiomics.shape[0] != x.shape[0]:
        raise ValueError(f"expected x[G,N,D] and radiomics[G,R], got {tuple(x.shape)} and {tuple(radiomics.shape)}")
    G, N, _ = x.shape
    return torch.cat([x, radiomics.to(x.dtype).unsqueeze(1).expand(G, N, radiomics.shape[1])], dim=2).contiguous()


def collect_teacher_outputs_device(model, bags, labels, image_ids, device, chunk=256, kmax=16):
    """`_collect_teacher_outputs` (`01_train_mil_teacher.py:69-87`) with every output left on the device.
    ``bags``: list of equal-sized [N, D] arrays or one [G, N, D] tensor."""
    dev = torch.device(device)
    model.eval()
    if isinstance(bags, torch.Tensor):
        x = bags.to(dev, torch.float32)
    else:
        x = torch.as_tensor(np.stack([np.asarray(b, dtype=np.float32) for b in bags])).to(dev)
    G, N, D = x.shape
    probs, att = [], []
    for lo in range(0, G, chunk):
        hi = min(G, lo + chunk)
        out = model(x[lo:hi].reshape(-1, D), BagOffsets.uniform(hi - lo, N, dev))
        probs.append(out["patch_probs"].view(hi - lo, N, -1))
        att.append(out["attention"].view(hi - lo, N))
    y = torch.as_tensor(np.asarray(labels, dtype=np.int64), device=dev)
    return DeviceTeacherOutputs(x, torch.cat(probs), torch.cat(att), y, image_ids, kmax=kmax)


def train_gnn_from_teacher(gnn_model, outputs_train, outputs_val, outputs_test, variant, *, fold=0, seed=42, **fit):
    """05's fold loop (`05_train_gnns.py:305-358`) fed directly from resident teacher outputs: no pickle, no upload."""
    from isic_hip import train as T
    recs = [o.graph_records(variant, fold, seed) for o in (outputs_train, outputs_val, outputs_test)]
    return T.train_gnn_fold(gnn_model, recs[0], recs[1], recs[2], **fit)
