"""Drop-in for the reference's ``02_compute_patch_statistics.py``: teacher outputs ->
patch statistics (adds ``dominant_class = argmax(patch_probs)``, reference `02:17`), keeping
the row schema (`02:19-26`).  Paths are arguments instead of hard-coded cluster paths and
the input is NOT deleted (the reference unlinks it, `02:38`)."""
import argparse
import pickle
from pathlib import Path

import numpy as np
import pandas as pd


def process_teacher_file(input_path, teacher_root, stats_root):
    input_path = Path(input_path)
    with open(input_path, "rb") as f:
        df = pickle.load(f)
    out = pd.DataFrame({
        "image_id": df["image_id"], "label": df["label"], "patch_embeddings": df["patch_embeddings"],
        "patch_probs": df["patch_probs"],
        "dominant_class": [np.argmax(p, axis=1) for p in df["patch_probs"]],
    })
    rel = input_path.relative_to(teacher_root)
    target = Path(str(Path(stats_root) / rel).replace("teacher_outputs_f", "patch_stats_f"))
    target.parent.mkdir(parents=True, exist_ok=True)
    with open(target, "wb") as f:
        pickle.dump(out, f)
    print(f"Saved: {target}")
    return target


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--teacher-outputs-root", default="teacher_outputs")
    ap.add_argument("--patch-stats-root", default="patch_stats")
    a = ap.parse_args()
    for p in sorted(Path(a.teacher_outputs_root).rglob("*.pkl")):
        if not p.name.startswith("patch_stats_"):
            process_teacher_file(p, a.teacher_outputs_root, a.patch_stats_root)
