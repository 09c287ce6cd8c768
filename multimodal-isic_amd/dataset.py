"""Drop-in for the reference's ``dataset.py`` dict contract (`dataset.py:42-56`:
keys image, mask, radiomics, age, sex, loc, artifacts, target, image_path,
segmentation_path) plus the synthetic ISIC-shaped bag datasets every BASELINE.json
config is measured on (there is no network for HAM10000 / ISIC images).

Image decoding itself (cv2 mask-centred crops, `dataset.py:58-99`) is CPU file I/O
outside the GPU path; ``DermDataset`` keeps it behind PIL so the module imports
without OpenCV."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

ARTIFACT_COLS = ['hair', 'ruler_marks', 'bubbles', 'vignette', 'frame', 'other']


class DermDataset(Dataset):
    def __init__(self, df, radiomics, transform=None, is_train=True, crop_size=450):
        self.df, self.radiomics = df, radiomics
        self.transform, self.is_train, self.crop_size = transform, is_train, crop_size
        self.artifact_cols = list(ARTIFACT_COLS)

    def __len__(self):
        return len(self.df)

    @staticmethod
    def _square_crop_on_mask(image, mask):
        """Largest centred-on-lesion square crop (`dataset.py:58-85` with crop = min(h, w))."""
        h, w = image.shape[:2]
        side = min(h, w)
        ys, xs = np.nonzero(mask)
        cx, cy = (int(xs.mean()), int(ys.mean())) if len(xs) else (w // 2, h // 2)
        x1 = min(max(cx - side // 2, 0), w - side)
        y1 = min(max(cy - side // 2, 0), h - side)
        return image[y1:y1 + side, x1:x1 + side], mask[y1:y1 + side, x1:x1 + side]

    def _load(self, image_path, mask_path):
        from PIL import Image
        image = np.asarray(Image.open(image_path).convert("RGB"))
        if mask_path == 'no_mask' or not os.path.exists(mask_path):
            mask = np.zeros(image.shape[:2], dtype=np.uint8)
        else:
            m = Image.open(mask_path).convert("L")
            if m.size != (image.shape[1], image.shape[0]):
                m = m.resize((image.shape[1], image.shape[0]), Image.NEAREST)
            mask = np.asarray(m)
        return self._square_crop_on_mask(image, mask)

    def __getitem__(self, idx):
        row = self.df.iloc[idx]
        image, mask = self._load(row['image_path'], row['segmentation_path'])
        if self.transform:
            aug = self.transform(image=image.astype(np.uint8), mask=mask.astype(np.uint8))
            image, mask = aug['image'], aug['mask']
        else:
            image = torch.from_numpy(np.ascontiguousarray(image)).permute(2, 0, 1).float() / 255.0
            mask = torch.from_numpy(np.ascontiguousarray(mask))[None].float() / 255.0
        has = lambda c: c in row.index
        return {
            'image': image, 'mask': mask,
            'radiomics': (torch.as_tensor(np.asarray(self.radiomics.iloc[idx].values, dtype=np.float32))
                          if self.radiomics is not None and hasattr(self.radiomics, "iloc")
                          else torch.zeros(102, dtype=torch.float)),          # reference stub, dataset.py:42
            'age': torch.tensor(row['age_normalized'] if has('age_normalized') else 0.0, dtype=torch.float),
            'sex': torch.tensor(row['sex_encoded'] if has('sex_encoded') else 0, dtype=torch.long),
            'loc': torch.tensor(row['loc_encoded'] if has('loc_encoded') else 0, dtype=torch.long),
            'artifacts': (torch.tensor(row[self.artifact_cols].values.astype(int), dtype=torch.long)
                          if all(has(c) for c in self.artifact_cols) else torch.zeros(6, dtype=torch.long)),
            'target': torch.tensor(row['dx'], dtype=torch.long),
            'image_path': row['image_path'], 'segmentation_path': row['segmentation_path'],
        }


def synthetic_latent_bags(n_bags, patches, dim, classes=7, shift=0.35, seed=42):
    """ISIC-shaped bags of patch latents (`01_train_mil_teacher.py:51-67` geometry: N x D per image):
    N(0,1) features with a planted class-dependent mean shift on a class-specific subset of patches, so
    that AUROC is non-trivial; balanced labels.  Returns (list of float32 arrays, labels)."""
    rng = np.random.RandomState(seed)
    labels = np.arange(n_bags) % classes
    rng.shuffle(labels)
    dirs = rng.randn(classes, dim).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    bags = []
    for y in labels:
        x = rng.randn(patches, dim).astype(np.float32)
        hot = rng.rand(patches) < 0.3
        x[hot] += shift * np.sqrt(dim) * 0.25 * dirs[y]
        bags.append(x)
    return bags, labels.astype(np.int64)


class SyntheticBagImages(Dataset):
    """BASELINE.json configs[1]: bags of K patches 3xSxS + an R-d radiomic vector, generated on the fly
    from a per-bag seed (deterministic, nothing stored)."""

    def __init__(self, n_bags=256, patches=64, size=224, radiomics_dim=128, classes=7, seed=42, shift=0.25):
        self.n, self.k, self.s, self.r, self.c, self.seed, self.shift = n_bags, patches, size, radiomics_dim, classes, seed, shift
        rng = np.random.RandomState(seed)
        self.labels = np.arange(n_bags) % classes
        rng.shuffle(self.labels)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        y = int(self.labels[i])
        img = torch.randn(self.k, 3, self.s, self.s, generator=g) + self.shift * (y - (self.c - 1) / 2)
        rad = torch.randn(self.r, generator=g) + self.shift * (y - (self.c - 1) / 2)
        return {'image': img, 'radiomics': rad, 'target': torch.tensor(y, dtype=torch.long)}
