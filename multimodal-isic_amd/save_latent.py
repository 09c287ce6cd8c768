"""Drop-in for the reference's ``save_latent.py`` (`extract_latents`, `:13-204`): frozen-encoder latents of every
image -> pooled / raw / patch-level DataFrames with the reference's columns.

What changes:
  * the encoder.  The reference runs an un-vendored ConvMAE conv-ViT (`save_latent.py:17-18,42-60`) whose code and
    weights are not in the tree; here the frozen encoder is the ResNet-18 of this build truncated after layer3
    (``ResNet18Encoder.run_tokens``): 14 x 14 = 196 tokens of 256 channels for a 224 x 224 image, one per 16 x 16-pixel
    patch -- the reference's token geometry (`:77`).  There is no random masking, so ``ids_keep = ids_restore =
    arange(196)`` (the reference passes ``mask_ratio=0``: every patch is kept, in shuffled order);
  * batched inference in HBM, the lesion-mask -> patch-flag step (`:73-87`) as one HIP launch
    (``isic_mask_patch_flags_f32``), and the per-patch Python double loop of ``build_patch_level_df`` (`:109-154`)
    replaced by array operations that produce the same rows in the same order.

``extract_latents(config, path, remove_background=False, datasets=None)`` keeps the reference's signature and return
tuple; ``datasets=(train_val_dataset, test_dataset)`` lets a caller hand in any dataset with the ``DermDataset`` dict
contract (`dataset.py:45-56`) -- the synthetic one below when there are no image files.
"""
from __future__ import annotations

import os

import numpy as np
import pandas as pd
import torch
from torch.utils.data import DataLoader, Dataset

from isic_hip.encoder import ResNet18Encoder
from isic_hip.lib import call

PATCH = 16                      # save_latent.py:77
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)       # save_latent.py:28


class SyntheticDermImages(Dataset):
    """ISIC-shaped stand-in with the ``DermDataset`` dict contract: a normalised 3x224x224 image, an elliptic lesion
    mask (some images without a mask), a class label; deterministic per index."""

    def __init__(self, n=32, size=224, classes=7, seed=42):
        self.n, self.size, self.classes, self.seed = n, size, classes, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 7919 + i)
        y = i % self.classes
        s = self.size
        img = torch.randn(3, s, s, generator=g) + 0.25 * (y - (self.classes - 1) / 2)
        yy, xx = torch.meshgrid(torch.arange(s), torch.arange(s), indexing="ij")
        cy, cx = (torch.rand(2, generator=g) * 0.5 + 0.25) * s
        ry, rx = (torch.rand(2, generator=g) * 0.25 + 0.08) * s
        mask = ((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1.0).float()
        if i % 5 == 4:
            mask = torch.zeros_like(mask)                       # the 'no_mask' case of dataset.py
        return {"image": img, "mask": mask.unsqueeze(0), "radiomics": torch.zeros(102), "age": torch.tensor(0.0),
                "sex": torch.tensor(0), "loc": torch.tensor(0), "artifacts": torch.zeros(6, dtype=torch.long),
                "target": torch.tensor(y, dtype=torch.long), "image_path": f"synthetic/img_{i:05d}.jpg",
                "segmentation_path": "no_mask" if i % 5 == 4 else f"synthetic/seg_{i:05d}.png"}


def mask_patch_flags(mask, patch=PATCH):
    """`save_latent.py:73-87` on the device: mask (B,H,W) or (B,1,H,W) -> bool [B, H/patch, W/patch]."""
    if mask.dim() == 4:
        mask = mask[:, 0]
    m = mask.contiguous().float()
    B, H, W = m.shape
    flags = torch.empty((B, H // patch, W // patch), device=m.device, dtype=torch.uint8)
    call("isic_mask_patch_flags_f32", m, flags, B, H, W, patch)
    return flags.bool()


def build_patch_level_df(latent_raw_df, remove=True):
    """`save_latent.py:109-158`: one row per kept patch -- image_path, segmentation_path, target, patch_id,
    patch_latent, patch_in_mask -- in image-major, token order; with ``remove`` only lesion-overlapping patches.
    Returns (frame, number of rows written while ``remove`` was on) exactly like the reference's counter."""
    cols = ["image_path", "segmentation_path", "target", "patch_id", "patch_latent", "patch_in_mask"]
    if len(latent_raw_df) == 0:
        return pd.DataFrame(columns=cols), 0
    lat = np.stack(latent_raw_df["latent"].values)                          # [B, T, D]
    ids = np.stack(latent_raw_df["ids_keep"].values).astype(np.int64)       # [B, T]
    flat = np.stack([np.asarray(m).ravel() for m in latent_raw_df["lesion_mask_patches"].values]).astype(bool)
    B, T, _ = lat.shape
    inside = np.take_along_axis(flat, np.minimum(ids, flat.shape[1] - 1), axis=1) & (ids < flat.shape[1])
    keep = inside if remove else np.ones_like(inside)
    b_idx, t_idx = np.nonzero(keep)
    df = pd.DataFrame({
        "image_path": latent_raw_df["image_path"].values[b_idx],
        "segmentation_path": latent_raw_df["segmentation_path"].values[b_idx],
        "target": latent_raw_df["target"].values[b_idx],
        "patch_id": ids[b_idx, t_idx],
        "patch_latent": list(lat[b_idx, t_idx]),
        "patch_in_mask": inside[b_idx, t_idx].astype(np.int64),
    }, columns=cols)
    return df, (int(keep.sum()) if remove else 0)


def _extract_from_loader(encoder, loader, device):
    pooled_list, raw_list = [], []
    for batch in loader:
        images = batch["image"].to(device)
        latent = encoder.run_tokens(images)                                 # [B, 196, 256] fp32, frozen / eval
        B, T, _ = latent.shape
        ids = np.tile(np.arange(T, dtype=np.int64), (B, 1))
        target = batch["target"].numpy()
        pooled_list.append(pd.DataFrame({
            "image_path": batch["image_path"], "segmentation_path": batch["segmentation_path"], "target": target,
            "latent_pooled_max": list(latent.max(dim=1).values.cpu().numpy()),          # :62
            "latent_pooled_mean": list(latent.mean(dim=1).cpu().numpy()),               # :63
            "ids_restore": list(ids), "ids_keep": list(ids)}))
        flags = mask_patch_flags(batch["mask"].to(device))
        raw_list.append(pd.DataFrame({
            "image_path": batch["image_path"], "segmentation_path": batch["segmentation_path"], "target": target,
            "latent": list(latent.cpu().numpy()), "ids_restore": list(ids), "ids_keep": list(ids),
            "lesion_mask_patches": list(flags.cpu().numpy())}))
    cat = lambda l: pd.concat(l, ignore_index=True) if l else pd.DataFrame()
    return cat(pooled_list), cat(raw_list)


def extract_latents(config, path, remove_background=False, datasets=None, batch_size=256):
    device = torch.device(config.get("device", "cuda:0"))
    seed = config.get("seed", 42)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if datasets is None:
        from dataset import DermDataset
        df_tv, df_te = pd.read_pickle(config["dir"]["df"]), pd.read_pickle(config["dir"]["df_test"])

        def transform(image, mask):                                          # A.Resize(224,224) + A.Normalize (:26-30)
            img = torch.from_numpy(np.ascontiguousarray(image)).permute(2, 0, 1).float().unsqueeze(0) / 255.0
            img = torch.nn.functional.interpolate(img, size=(224, 224), mode="bilinear", align_corners=False)[0]
            img = (img - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
            m = torch.from_numpy(np.ascontiguousarray(mask)).float()[None, None]
            m = torch.nn.functional.interpolate(m, size=(224, 224), mode="nearest")[0, 0]
            return {"image": img, "mask": m}
        datasets = (DermDataset(df_tv, radiomics=None, transform=transform), DermDataset(df_te, radiomics=None, transform=transform))
    loaders = [DataLoader(d, batch_size=batch_size, shuffle=False) for d in datasets]
    if str(config.get("encoder", "resnet18")).lower() in ("vit_s16", "vit-s/16", "vit_small_patch16_224"):
        from isic_hip.vit import ViTSmallEncoder                # BASELINE.json configs[4]: ViT-S/16, fp16, 196 x 384 tokens
        enc = ViTSmallEncoder().to(device)
    else:
        enc = ResNet18Encoder().to(device)
    ckpt = os.path.join(os.getcwd(), config.get("model_path", "models"), path)
    if os.path.exists(ckpt):
        enc.load_state_dict(torch.load(ckpt, map_location=device, weights_only=True), strict=False)          # :46-48
    else:
        print(f"save_latent: checkpoint {ckpt} not found -- encoder keeps its seeded initialisation")
    enc.eval()
    latent_pooled_train, latent_raw_train = _extract_from_loader(enc, loaders[0], device)
    latent_pooled_test, latent_raw_test = _extract_from_loader(enc, loaders[1], device)
    patch_level_train_df, train_count = build_patch_level_df(latent_raw_train, remove=remove_background)
    patch_level_test_df, test_count = build_patch_level_df(latent_raw_test, remove=remove_background)
    print(f"Total lesion-overlapping patches (train_val): {train_count}")
    print(f"Total lesion-overlapping patches (test): {test_count}")
    if bool(config.get("pca", False)):                                       # :163-180
        from sklearn.decomposition import PCA
        if len(patch_level_train_df) > 0:
            Xtr = np.vstack(patch_level_train_df["patch_latent"].values)
            pca = PCA(n_components=0.90, whiten=False)
            Xp = pca.fit_transform(Xtr)
            print(f"PCA reduced dimensions from {Xtr.shape[1]} to {Xp.shape[1]}")
            patch_level_train_df["patch_latent_pca"] = list(Xp)
        else:
            patch_level_train_df["patch_latent_pca"] = []
        if len(patch_level_test_df) > 0:
            if len(patch_level_train_df) == 0:
                raise RuntimeError("No train patches to fit PCA. Cannot transform test patches.")
            patch_level_test_df["patch_latent_pca"] = list(pca.transform(np.vstack(patch_level_test_df["patch_latent"].values)))
        else:
            patch_level_test_df["patch_latent_pca"] = []
    else:
        print("PCA disabled via config; using raw patch_latent as patch_latent_pca.")
        patch_level_train_df["patch_latent_pca"] = patch_level_train_df["patch_latent"]
        patch_level_test_df["patch_latent_pca"] = patch_level_test_df["patch_latent"]
    os.makedirs("dataframes_latents", exist_ok=True)                         # :188-190 (save_files = False)
    print("Finished saving train_val and test patch-level and pooled latents.")
    return (patch_level_train_df, patch_level_test_df, latent_pooled_train, latent_pooled_test, latent_raw_train,
            latent_raw_test)
