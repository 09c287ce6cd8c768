"""Cross-validated patch-graph classification on the MI355X: drop-in for the reference's
``05_train_gnns.py`` -- same inputs (``patch_stats/<model>/patch_stats_fold_<f>_<split>.pkl``
+ ``graph_outputs/<model>/graph_dataset.pkl``), same CLI flags, same resumable
``results_job_<id>.csv`` / ``detailed_fold_results_job_<id>.csv`` bookkeeping -- with the
GraphMIL forward/backward on HIP kernels, ``--graphs-per-step`` graphs per optimizer step in one
launch, and one-process-per-GPU data parallelism under ``torch.distributed.run``.
"""
import argparse
import copy
import os
import pickle
import random
import sys
from pathlib import Path

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import pandas as pd
import torch

from gnn_models import GNN_TYPES, GraphMIL  # noqa: F401  (GraphMIL re-exported like the reference module)
from isic_hip import train as T

DEFAULT_NEIGHBORS = tuple(range(1, 9)) + (12, 16)
RESULT_KEY = ["embedding_model", "graph_variant", "graph_model", "seed", "hidden_dim", "num_layers", "dropout",
              "learning_rate", "weight_decay"]
METRICS = ("accuracy", "bacc", "auc", "macro_f1")


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def graph_variants():
    return ["grid4", "grid8"] + [f"knn{k}" for k in DEFAULT_NEIGHBORS] + [f"random{r}" for r in DEFAULT_NEIGHBORS]


def edge_index_for_variant(row, variant):
    """`05:228-239`."""
    if variant == "none":
        return None
    if variant in ("grid4", "grid8"):
        return np.asarray(row[f"{variant}_edge_index"], dtype=np.int64)
    for prefix, col in (("knn", "knn_edge_indices"), ("random", "random_edge_indices")):
        if variant.startswith(prefix):
            return np.asarray(row[col][int(variant[len(prefix):])], dtype=np.int64)
    raise ValueError(f"Unknown graph variant: {variant}")


def _frame(path):
    with Path(path).open("rb") as fh:
        obj = pickle.load(fh)
    return obj if isinstance(obj, pd.DataFrame) else pd.DataFrame(obj)


def load_fold_records(root, embedding_model, fold, split, variant):
    """`05:248-270`: graph blueprints joined one-to-one with patch statistics."""
    gpath = Path(root) / "graph_outputs" / embedding_model / "graph_dataset.pkl"
    spath = Path(root) / "patch_stats" / embedding_model / f"patch_stats_fold_{fold}_{split}.pkl"
    if not gpath.exists() or not spath.exists():
        raise FileNotFoundError(f"Missing graph or patch stats for {embedding_model}, fold {fold}, {split}")
    graphs = _frame(gpath)
    graphs = graphs[(graphs["fold"] == fold) & (graphs["split"] == split)]
    stats = _frame(spath)[["image_id", "label", "patch_embeddings"]]
    merged = graphs.merge(stats, on="image_id", how="inner", validate="one_to_one")
    if len(merged) != len(graphs):
        raise ValueError(f"{embedding_model}, fold {fold}, {split}: graph rows without patch statistics")
    records = []
    for _, row in merged.iterrows():
        x = np.asarray(row["patch_embeddings"], dtype=np.float32)
        ei = edge_index_for_variant(row, variant)
        if x.ndim != 2 or ei.shape[0] != 2:
            raise ValueError(f"Invalid graph record for image {row['image_id']}")
        records.append({"x": x, "edge_index": ei, "y": int(row["label"]), "image_id": str(row["image_id"])})
    return records


def train_one_fold(train_records, val_records, test_records, args, fold, num_classes, input_dim, device):
    """`05:305-358` (model configuration of the reference call site, `05:310-326`)."""
    set_seed(args.seed + fold)
    model = GraphMIL(input_dim=input_dim, gnn_type=args.gnn if isinstance(args.gnn, str) else args.gnn[0],
                     gnn_hidden=args.hidden_dim, gnn_layers=args.num_layers, gnn_dropout=args.dropout, gnn_heads=4,
                     gnn_concat=True, att_dim=128, att_heads=4, pool_dropout=0.2, classifier_dim=128,
                     classifier_light=True, num_classes=num_classes, use_residual=True, use_layer_norm=True).to(device)
    model.set_dropout_state(args.seed + fold, 0)
    return T.train_gnn_fold(model, train_records, val_records, test_records, lr=args.learning_rate,
                            weight_decay=args.weight_decay, epochs=args.epochs, patience=args.patience,
                            min_delta=args.min_delta, graphs_per_step=args.graphs_per_step, num_classes=num_classes,
                            device=device)


def aggregate(rows, prefix):
    out = {}
    for m in METRICS:
        vals = [r[m] for r in rows]
        out[f"{prefix}_{m}_mean"] = float(np.nanmean(vals))
        out[f"{prefix}_{m}_std"] = float(np.nanstd(vals, ddof=0))
    return out


def export_detailed_fold_results(fold_records, output_path):
    output_path.parent.mkdir(parents=True, exist_ok=True)
    new = pd.DataFrame(fold_records)
    if output_path.exists():
        new = pd.concat([pd.read_csv(output_path), new], ignore_index=True).drop_duplicates(
            subset=["embedding_model", "graph_variant", "graph_model", "fold", "seed"], keep="last")
    new.to_csv(output_path, index=False)


def save_results(results, output_path):
    """Atomic replace so finished variants survive a job timeout (`05:386-393`)."""
    output_path.parent.mkdir(parents=True, exist_ok=True)
    tmp = output_path.with_suffix(output_path.suffix + ".tmp")
    results.sort_values(["embedding_model", "graph_variant", "graph_model"]).to_csv(tmp, index=False)
    os.replace(tmp, output_path)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__)
    p.add_argument("--root", type=Path, default=Path(__file__).resolve().parent)
    p.add_argument("--models", nargs="*")
    p.add_argument("--variants", nargs="*", default=graph_variants())
    p.add_argument("--folds", nargs="*", type=int, default=list(range(1)))
    p.add_argument("--gnn", nargs="+", choices=GNN_TYPES, default=["mlp", "gcn", "gcnii", "graphsage", "gin", "gat"])
    p.add_argument("--epochs", type=int, default=1)
    p.add_argument("--patience", type=int, default=16)
    p.add_argument("--min-delta", type=float, default=1e-6)
    p.add_argument("--hidden-dim", type=int, default=128)
    p.add_argument("--num-layers", type=int, default=2)
    p.add_argument("--dropout", type=float, default=0.5)
    p.add_argument("--learning-rate", type=float, default=1e-4)
    p.add_argument("--weight-decay", type=float, default=1e-4)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--device", default="cuda")
    p.add_argument("--results-csv", type=Path, default=Path("gnn_results/common_results.csv"))
    p.add_argument("--job-id", type=str, default="0")
    p.add_argument("--graphs-per-step", type=int, default=1, help="graphs per optimizer step per GPU (1 = reference)")
    return p.parse_args(argv)


def run_gnn_experiments(args):
    rank, world = T.dist_info()
    root = args.root.resolve()
    device = torch.device("cuda", torch.cuda.current_device())
    available = sorted(p.name for p in (root / "graph_outputs").iterdir() if p.is_dir())
    models = args.models or available
    unknown = sorted(set(args.variants) - set(graph_variants()))
    if unknown:
        raise ValueError(f"Unsupported variants: {unknown}")
    out_dir = args.results_csv.parent if args.results_csv.is_absolute() else root / args.results_csv.parent
    out_dir.mkdir(parents=True, exist_ok=True)
    out_path = out_dir / f"results_job_{args.job_id}.csv"
    detail_path = out_dir / f"detailed_fold_results_job_{args.job_id}.csv"
    done = set()
    for csv in out_dir.glob("results_job_*.csv"):
        try:
            prev = pd.read_csv(csv)
            if set(RESULT_KEY).issubset(prev.columns):
                done.update(map(tuple, prev[RESULT_KEY].itertuples(index=False, name=None)))
        except Exception:
            pass
    results = pd.read_csv(out_path) if out_path.exists() else pd.DataFrame()
    for emb in models:
        if emb not in available:
            raise FileNotFoundError(f"No graph artifacts for embedding model {emb}")
        for variant in (["none"] if args.gnn.lower() == "mlp" else args.variants):
            key = (emb, variant, args.gnn, args.seed, args.hidden_dim, args.num_layers, args.dropout,
                   args.learning_rate, args.weight_decay)
            if key in done:
                if rank == 0:
                    print(f"Skipping completed experiment: {emb} | {variant} | {args.gnn}")
                continue
            load_variant = "grid4" if variant == "none" else variant
            vals, tests, epochs, detail = [], [], [], []
            for fold in args.folds:
                tr, va, te = (load_fold_records(root, emb, fold, s, load_variant) for s in ("train", "val", "test"))
                n_cls = max(r["y"] for r in tr + va + te) + 1
                vm, tm, be = train_one_fold(tr, va, te, args, fold, n_cls, tr[0]["x"].shape[1], device)
                vals.append(vm); tests.append(tm); epochs.append(be)
                detail.append({"embedding_model": emb, "graph_variant": variant, "graph_model": args.gnn, "fold": fold,
                               "seed": args.seed, "hidden_dim": args.hidden_dim, "num_layers": args.num_layers,
                               "dropout": args.dropout, "best_epoch": be,
                               **{f"test_{k}": v for k, v in tm.items()}, **{f"val_{k}": v for k, v in vm.items()}})
                if rank == 0:
                    print(f"{emb} | {variant} | fold {fold}: val BAcc={vm['bacc']:.4f}, test BAcc={tm['bacc']:.4f}")
            row = {"embedding_model": emb, "graph_variant": variant, "graph_model": args.gnn, "num_folds": len(args.folds),
                   "seed": args.seed, "hidden_dim": args.hidden_dim, "num_layers": args.num_layers, "dropout": args.dropout,
                   "learning_rate": args.learning_rate, "weight_decay": args.weight_decay,
                   "best_epoch_mean": float(np.mean(epochs)), "best_epoch_std": float(np.std(epochs, ddof=0)),
                   **aggregate(vals, "val"), **aggregate(tests, "test")}
            results = pd.concat([results, pd.DataFrame([row])], ignore_index=True).drop_duplicates(RESULT_KEY, keep="last")
            if rank == 0:
                save_results(results, out_path)
                export_detailed_fold_results(detail, detail_path)
                print(f"Saved {len(results)} completed experiment rows to {out_path}")
            done.add(key)


def main(argv=None):
    base = parse_args(argv)
    T.init_distributed()
    for gnn_type in base.gnn:
        args = copy.copy(base)
        args.gnn = gnn_type
        run_gnn_experiments(args)


if __name__ == "__main__":
    main()
