"""Evaluation metrics restated in numpy (oracle / test infrastructure).

The reference uses scikit-learn: macro one-vs-rest AUROC
(`05_train_gnns.py:290`, `01_train_mil_teacher.py:266`) and balanced accuracy
(`05_train_gnns.py:299`, `01_train_mil_teacher.py:269`).  Pinned by
``tests/golden/metrics.npz`` (sklearn 1.7.2 outputs on a fixed score matrix).
"""
from __future__ import annotations

import numpy as np


def _average_ranks(v):
    order = np.argsort(v, kind="mergesort")
    sv = v[order]
    ranks = np.empty(len(v), dtype=np.float64)
    i = 0
    while i < len(v):
        j = i
        while j + 1 < len(v) and sv[j + 1] == sv[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return ranks


def binary_auc(y, s):
    y = np.asarray(y).astype(bool)
    npos, nneg = int(y.sum()), int((~y).sum())
    if npos == 0 or nneg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    r = _average_ranks(np.asarray(s, dtype=np.float64))
    return (r[y].sum() - npos * (npos + 1) / 2.0) / (npos * nneg)


def roc_auc_ovr_macro(y_true, scores, num_classes=None):
    """``roc_auc_score(y, scores, multi_class='ovr', labels=arange(C))`` (macro)."""
    y_true = np.asarray(y_true)
    scores = np.asarray(scores, dtype=np.float64)
    c = scores.shape[1] if num_classes is None else num_classes
    return float(np.mean([binary_auc(y_true == k, scores[:, k]) for k in range(c)]))


def balanced_accuracy(y_true, y_pred):
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    return float(np.mean([np.mean(y_pred[y_true == k] == k) for k in np.unique(y_true)]))
