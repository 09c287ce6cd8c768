"""Drop-in for the numeric part of the reference's ``04_measure_heterophily.py``: the master summary of the
heterophily measures over all images, folds and graph variants (`build_master_summary`, 04:188-226), computed on
the MI355X (``measure_heterophily.py``).  The reference's plotting / aggregation section (04:229-589) and its
hard-coded cluster paths (04:23,560-561) are out of scope: paths are flags here and the output is one CSV.

    python 04_measure_heterophily.py --graph-outputs-root graph_outputs --patch-stats-root patch_stats --out-csv h.csv
"""
import argparse
import os
import pickle

import numpy as np
import pandas as pd

import measure_heterophily as mh


def build_master_summary(graph_root, patch_root, images_per_launch=64, device="cuda:0"):
    rows = []
    for model in sorted(os.listdir(graph_root)):
        gpath = os.path.join(graph_root, model, "graph_dataset.pkl")
        if not os.path.exists(gpath):
            continue
        gdf = pd.DataFrame(pickle.load(open(gpath, "rb")))
        for (fold, split), grp in gdf.groupby(["fold", "split"]):
            ppath = os.path.join(patch_root, model, f"patch_stats_fold_{int(fold)}_{split}.pkl")
            pdf = pd.DataFrame(pickle.load(open(ppath, "rb")))
            merged = grp.merge(pdf, on="image_id", how="inner", suffixes=("", "_patch"))        # 04:80-85
            if merged.empty:
                continue
            first = merged.iloc[0]
            variants = ["grid4", "grid8"] + [f"knn{int(k)}" for k in first["knn_edge_indices"]] + \
                       [f"random{int(r)}" for r in first["random_edge_indices"]]
            for variant in variants:
                for lo in range(0, len(merged), images_per_launch):
                    chunk = merged.iloc[lo:lo + images_per_launch]
                    recs = chunk.to_dict("records")
                    ems = mh.compute_edge_heterophily_batch(
                        [r["patch_embeddings"] for r in recs], [r["patch_probs"] for r in recs],
                        [r["dominant_class"] for r in recs], [mh.edge_index_from_variant(r, variant) for r in recs],
                        device=device)
                    for r, em in zip(recs, ems):
                        kind = "grid" if variant.startswith("grid") else ("knn" if variant.startswith("knn") else "random")
                        meta = {"model_name": r.get("model_name", model), "fold": int(fold), "split": split,
                                "image_id": r["image_id"], "label": r.get("label"), "graph_variant": variant,
                                "graph_type": kind,
                                "graph_param": None if kind == "grid" else int(variant[len(kind):])}      # 04:211-222
                        rows.append(mh.summarize_image(em, meta))
    return pd.DataFrame.from_records(rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph-outputs-root", default="graph_outputs")
    ap.add_argument("--patch-stats-root", default="patch_stats")
    ap.add_argument("--out-csv", default="heterophily_summary.csv")
    ap.add_argument("--images-per-launch", type=int, default=64)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()
    df = build_master_summary(a.graph_outputs_root, a.patch_stats_root, a.images_per_launch, a.device)
    df.drop(columns=["H_compat_matrix"]).to_csv(a.out_csv, index=False)
    print(f"wrote {len(df)} rows ({df['graph_variant'].nunique() if len(df) else 0} graph variants) to {a.out_csv}")


if __name__ == "__main__":
    main()
