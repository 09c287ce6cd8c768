"""Latent extraction bookkeeping restated on CPU (oracle / test infrastructure): the lesion-mask -> patch flags of
`save_latent.py:73-87` (torch unfold, as the reference writes it) and ``build_patch_level_df`` (`:109-158`) as the
reference's per-patch Python double loop.  PARITY UNPINNED by fixtures: both are closures inside ``extract_latents``,
which cannot run here (its encoder, ConvMAE, is un-vendored; albumentations is absent) -- the restatement follows the
source line by line instead."""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch


def mask_patch_flags(mask, patch_size=16):
    if mask.dim() == 3:
        mask = mask.unsqueeze(1)                                                       # :74-75
    B, _, H, W = mask.shape
    mp = mask.unfold(2, patch_size, patch_size).unfold(3, patch_size, patch_size)      # :80
    mp = mp.contiguous().view(B, 1, H // patch_size, W // patch_size, -1).sum(dim=-1)  # :81-82
    return (mp > 0).squeeze(1)                                                         # :83


def build_patch_level_df(latent_raw_df, remove=True):
    rows, count = [], 0
    for _, row in latent_raw_df.iterrows():                                            # :112
        mask_flat = np.asarray(row["lesion_mask_patches"]).ravel()                     # :121
        for patch_latent, patch_id in zip(row["latent"], row["ids_keep"]):             # :123
            patch_id = int(patch_id)
            inside = False
            patch_in_mask = 0
            if patch_id < mask_flat.size:                                              # :129-131
                inside = bool(mask_flat[patch_id])
                patch_in_mask = int(inside)
            rec = {"image_path": row["image_path"], "segmentation_path": row["segmentation_path"], "target": row["target"],
                   "patch_id": patch_id, "patch_latent": np.asarray(patch_latent), "patch_in_mask": patch_in_mask}
            if remove:
                if inside:                                                             # :133-143
                    rows.append(rec)
                    count += 1
            else:
                rows.append(rec)                                                       # :144-152
    return pd.DataFrame(rows), count
