"""MIL-teacher training on the MI355X: drop-in for the reference's
``01_train_mil_teacher.py`` (5-fold stratified CV over patient bags, class-balanced
sampling, Adam/AdamW from ``config['best_params']``, early stopping on validation loss,
``teacher_outputs/<ckpt>/teacher_outputs_fold_{f}_{split}.pkl`` with the reference's row
schema) -- plus several bags per optimizer step and one-process-per-GPU data parallelism:

    python 01_train_mil_teacher.py --config_path config.yml --synthetic
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        01_train_mil_teacher.py --synthetic --bags-per-step 8

Inputs: the pickled patch-level latent frames of the reference's ``load`` branch
(`01_train_mil_teacher.py:146-152`; columns image_path, patch_latent, target[, image_id,
patch_id]) named in ``config['dir']`` or by ``--patch-train-df/--patch-test-df``, or
``--synthetic`` ISIC-shaped latents (196 x 768 per image).  The ConvMAE latent extraction
itself (`save_latent.py`) is outside this build (encoder un-vendored).
"""
import argparse
import os
import pickle
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import pandas as pd
import torch
import yaml
from sklearn.metrics import accuracy_score, precision_recall_fscore_support, roc_auc_score
from sklearn.model_selection import StratifiedKFold

from dataset import synthetic_latent_bags
from isic_hip import train as T
from utils import get_args_parser
from utils_g_mil import AttentionMIL_teacher

SPLITS, SEED, NUM_EPOCHS = 5, 42, 200


def _patch_order(group):
    """Patches of one image in patch-id order (`01:19-49`): explicit patch_id, else the trailing integer
    of the file name, else as stored."""
    if "patch_id" in group.columns:
        return group.sort_values("patch_id")
    num = group["image_path"].map(lambda p: os.path.splitext(os.path.basename(p))[0].split("_")[-1])
    if num.str.fullmatch(r"\d+").all():
        return group.assign(_n=num.astype(int)).sort_values("_n").drop(columns="_n")
    return group


def build_patient_bags(df):
    """`01:51-67`: one bag per patient_id -> (features [N,D], majority label, image id)."""
    feats, labels, ids = [], [], []
    for _, grp in df.groupby("patient_id"):
        grp = _patch_order(grp)
        feats.append(np.vstack(grp["patch_latent"].values).astype(np.float32))
        labels.append(int(grp["target"].mode().iat[0]))
        ids.append(str(grp["image_id"].iloc[0]) if "image_id" in grp.columns
                   else os.path.splitext(os.path.basename(grp["image_path"].iloc[0]))[0])
    return feats, labels, ids


def load_bags(config, args):
    if args.synthetic:
        syn = config.get("synthetic", {})
        n = args.synthetic_bags or int(syn.get("bags", 256))
        bags, labels = synthetic_latent_bags(n, int(syn.get("patches_per_bag", 196)), int(syn.get("latent_dim", 768)),
                                             int(syn.get("classes", 7)), float(syn.get("class_shift", 0.35)), SEED)
        cut = int(0.8 * n)
        ids = [f"SYN_{i:07d}" for i in range(n)]
        return (bags[:cut], labels[:cut].tolist(), ids[:cut]), (bags[cut:], labels[cut:].tolist(), ids[cut:])
    out = []
    for key, flag in (("patch_train_df", args.patch_train_df), ("patch_test_df", args.patch_test_df)):
        path = flag or config.get("dir", {}).get(key)
        if not path:
            raise FileNotFoundError(f"no {key}: pass --{key.replace('_', '-')} / set dir.{key}, or use --synthetic")
        with open(path, "rb") as f:
            df = pickle.load(f)
        df["patient_id"] = df["image_path"].apply(lambda x: os.path.basename(x).split("_")[1].split(".")[0])  # 01:158
        out.append(build_patient_bags(df))
    return out[0], out[1]


def test_metrics(model, state, bags, labels, device):
    """`01:89-118`."""
    if state is not None:
        model.load_state_dict(state)
    if not labels:
        return {k: np.nan for k in ("micro", "macro_p", "macro_r", "macro_f1", "weighted_p", "weighted_r", "weighted_f1")}
    probs, _ = T.eval_teacher(model, T.BagStore(bags, device), labels)
    y, pred = np.asarray(labels), probs.argmax(axis=1)
    mp, mr, mf, _ = precision_recall_fscore_support(y, pred, average="macro", zero_division=0)
    wp, wr, wf, _ = precision_recall_fscore_support(y, pred, average="weighted", zero_division=0)
    return {"micro": accuracy_score(y, pred), "macro_p": mp, "macro_r": mr, "macro_f1": mf,
            "weighted_p": wp, "weighted_r": wr, "weighted_f1": wf}


def _auc(y, s):
    try:
        return {"val_auc": roc_auc_score(y, s, multi_class="ovr")}
    except Exception:
        return {}


def main():
    parser = get_args_parser("config.yml")
    parser.add_argument("--synthetic", action="store_true")
    parser.add_argument("--synthetic-bags", type=int, default=0)
    parser.add_argument("--patch-train-df", default="")
    parser.add_argument("--patch-test-df", default="")
    parser.add_argument("--bags-per-step", type=int, default=1, help="bags per optimizer step per GPU (1 = reference)")
    parser.add_argument("--epochs", type=int, default=NUM_EPOCHS)
    parser.add_argument("--folds", type=int, default=SPLITS)
    parser.add_argument("--model-name", default="synthetic.pth")
    parser.add_argument("--out-dir", default="teacher_outputs")
    args, _ = parser.parse_known_args()
    cfg_path = args.config_path if os.path.exists(args.config_path) else os.path.join(os.path.dirname(__file__), args.config_path)
    with open(cfg_path) as f:
        config = yaml.load(f, Loader=yaml.SafeLoader)
    rank, world = T.init_distributed()
    device = torch.device("cuda", torch.cuda.current_device())
    (tr_f, tr_y, tr_id), (te_f, te_y, te_id) = load_bags(config, args)
    bp = config["best_params"]
    out_dir = os.path.join(args.out_dir, os.path.splitext(args.model_name)[0])
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    skf = StratifiedKFold(n_splits=args.folds, shuffle=True, random_state=SEED)
    y_all = np.asarray(tr_y)
    for fold, (tri, vai) in enumerate(skf.split(np.zeros(len(y_all)), y_all)):
        if rank == 0:
            print(f"  Model {args.model_name} | Fold {fold + 1}/{args.folds}: train {len(tri)} patients, val {len(vai)} patients")
        f_tr, y_tr, id_tr = [tr_f[i] for i in tri], [int(tr_y[i]) for i in tri], [tr_id[i] for i in tri]
        f_va, y_va, id_va = [tr_f[i] for i in vai], [int(tr_y[i]) for i in vai], [tr_id[i] for i in vai]
        for seeder in (torch.manual_seed, np.random.seed, random.seed, torch.cuda.manual_seed_all):   # 01:203-207
            seeder(SEED + fold)
        model = AttentionMIL_teacher(input_dim=f_tr[0].shape[1], hidden_dim=bp["hidden_dim"], att_dim=bp["att_dim"],
                                     dropout=bp["dropout"], num_classes=len(set(y_tr))).to(device)
        model.set_dropout_state(SEED + fold, 0)
        patience = config.get("training_plan", {}).get("parameters", {}).get("patience", 8)
        res = T.train_teacher_fold(model, f_tr, y_tr, f_va, y_va, optimizer=bp["optimizer"], lr=float(bp["learning_rate"]),
                                   weight_decay=float(bp["weight_decay"]), epochs=args.epochs, patience=patience,
                                   bags_per_step=args.bags_per_step, seed=SEED + fold, device=device, metric_fn=_auc)
        m_bacc = test_metrics(model, res["best_state_bacc"], te_f, te_y, device)
        m_loss = test_metrics(model, res["best_state_loss"] or res["best_state_bacc"], te_f, te_y, device)
        if rank == 0:
            print(f"    test (best-bacc state): {m_bacc}\n    test (best-loss state): {m_loss}")
            if res["best_state_bacc"] is not None:
                model.load_state_dict(res["best_state_bacc"])
            for split, (f_, y_, id_) in (("train", (f_tr, y_tr, id_tr)), ("val", (f_va, y_va, id_va)), ("test", (te_f, te_y, te_id))):
                T.collect_teacher_outputs(model, f_, y_, id_, device).to_pickle(
                    os.path.join(out_dir, f"teacher_outputs_fold_{fold}_{split}.pkl"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
