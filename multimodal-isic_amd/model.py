"""MI355X-native drop-in for the reference's ``model.py`` surface.

* ``AttentionFusion`` / ``AttentionFusion_Late`` / ``MultiModalFusionNet`` keep the
  reference constructors, ``state_dict`` names and ``forward`` keywords
  (reference `model.py:6-227`); their arithmetic (radiomics / clinical / artifact
  MLPs ``Linear->LayerNorm->ReLU->Dropout`` x2, attention fusion, fusion head) runs
  on the HIP kernels of ``libisic_hip.so``.  The reference's EfficientNet-b3 image
  branch needs a network fetch (`model.py:58`) and is not on the MIL/GNN path; the
  ``image`` modality here is served by the ResNet-18 patch encoder instead.
* ``MultiModalMILNet`` is the model ``BASELINE.json`` configs[1] describes and the
  reference does not contain: bags of K image patches -> ResNet-18 patch encoder ->
  attention-MIL teacher head (`utils_g_mil.py:38-105`) -> bag embedding, fused with
  the radiomic MLP (`model.py:74-83`) through the reference's intermediate fusion
  (`model.py:206-216`).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from isic_hip import ops
from isic_hip.bags import BagOffsets, as_offsets
from isic_hip.encoder import LAYERS, ResNet18Encoder
from utils_g_mil import AttentionMIL_teacher, _DropoutClock


def _mlp_ln(in_dim, mid, out, p1, p2):
    # reference layout `model.py:63-104`: indices 0,1 (Linear, LayerNorm) and 4,5
    return nn.Sequential(nn.Linear(in_dim, mid), nn.LayerNorm(mid), nn.ReLU(), nn.Dropout(p1),
                         nn.Linear(mid, out), nn.LayerNorm(out), nn.ReLU(), nn.Dropout(p2))


def _run_mlp_ln(seq, x, clock, site, training):
    """Linear -> LayerNorm -> ReLU -> Dropout, twice (`model.py:74-83`), on HIP."""
    h = ops.linear(x, seq[0].weight, seq[0].bias)
    h = ops.layer_norm(h, seq[1].weight, seq[1].bias, seq[1].eps, relu=True,
                       drop=clock.spec(seq[3].p, site, training))
    h = ops.linear(h, seq[4].weight, seq[4].bias)
    return ops.layer_norm(h, seq[5].weight, seq[5].bias, seq[5].eps, relu=True,
                          drop=clock.spec(seq[7].p, site + 1, training))


def _run_fusion_mlp(seq, x, clock, site, training):
    """Linear -> ReLU -> Dropout -> Linear (`model.py:129-143`)."""
    h = ops.linear(x, seq[0].weight, seq[0].bias, ops.ACT_RELU, clock.spec(seq[2].p, site, training))
    return ops.linear(h, seq[3].weight, seq[3].bias)


class AttentionFusion(nn.Module):
    """Reference `model.py:6-23`: softmax-over-modalities attention pooling of M feature
    vectors -- the same segmented attention pool as the MIL head with bags of M."""

    def __init__(self, input_dim, num_modalities):
        super().__init__()
        self.attn = nn.Sequential(nn.Linear(input_dim, 128), nn.Tanh(), nn.Linear(128, 1))

    def forward(self, features):
        stacked = torch.stack(features, dim=1)                    # [B, M, D]
        B, M, D = stacked.shape
        offs = BagOffsets.uniform(B, M, stacked.device)
        z, _ = ops.attn_pool(stacked.reshape(B * M, D), self.attn[0].weight, self.attn[0].bias, self.attn[2].weight,
                             self.attn[2].bias, offs.device, offs.max_bag, heads=1)
        return z


class AttentionFusion_Late(nn.Module):
    """Reference `model.py:25-40`."""

    def __init__(self, num_modalities, num_classes=7):
        super().__init__()
        self.attention_net = nn.Sequential(nn.Linear(num_modalities * num_classes, 128), nn.ReLU(),
                                           nn.Linear(128, num_modalities))

    def forward(self, logits):
        cat = torch.cat(logits, dim=1)
        s = ops.linear(ops.linear(cat, self.attention_net[0].weight, self.attention_net[0].bias, ops.ACT_RELU),
                       self.attention_net[2].weight, self.attention_net[2].bias)
        w = ops.softmax_rows(s).unsqueeze(2)
        return (torch.stack(logits, dim=1) * w).sum(dim=1)


class MultiModalFusionNet(nn.Module):
    """Reference `model.py:42-227` (same constructor / forward keywords)."""

    def __init__(self, modality=['image', 'radiomics', 'clinical', 'artifacts'], fusion_level='intermediate',
                 fusion_strategy='attention', radiomics_dim=780, num_sex_classes=3, num_loc_classes=15,
                 num_artifact_classes=6, num_classes=7):
        super().__init__()
        self.modality, self.fusion_level, self.fusion_strategy = list(modality), fusion_level, fusion_strategy
        self.shared_dim = 128
        if 'image' in self.modality:
            self.image_model = ResNet18Encoder()
            self.image_proj = _mlp_ln(self.image_model.out_dim, 256, 128, 0.3, 0.2)
        self.radiomics_mlp = _mlp_ln(radiomics_dim, 256, 128, 0.4, 0.3)
        self.clinical_mlp = _mlp_ln(13, 64, 128, 0.2, 0.2)
        self.artifact_mlp = _mlp_ln(12, 64, 128, 0.2, 0.2)
        self.sex_emb = nn.Embedding(num_sex_classes, 4)
        self.loc_emb = nn.Embedding(num_loc_classes, 8)
        self.artifact_embeddings = nn.ModuleList([nn.Embedding(2, 2) for _ in range(num_artifact_classes)])
        self.feature_dims = [128 for _ in self.modality]
        self.total_dim = sum(self.feature_dims)
        if fusion_level == 'intermediate':
            if fusion_strategy in ('concat', 'weighted'):
                self.fusion_mlp = nn.Sequential(nn.Linear(self.total_dim, 256), nn.ReLU(), nn.Dropout(0.4),
                                                nn.Linear(256, num_classes))
            elif fusion_strategy == 'attention':
                self.attention = AttentionFusion(128, len(self.modality))
                self.fusion_mlp = nn.Sequential(nn.Linear(128, 256), nn.ReLU(), nn.Dropout(0.4),
                                                nn.Linear(256, num_classes))
            else:
                raise ValueError(f"Unknown fusion_strategy: {fusion_strategy}")
            if fusion_strategy == 'weighted':
                self.weights = nn.Parameter(torch.ones(len(self.modality)) / len(self.modality))
        elif fusion_level == 'late':
            if fusion_strategy == 'weighted':
                self.weights = nn.Parameter(torch.ones(len(self.modality)) / len(self.modality))
            elif fusion_strategy == 'attention':
                self.attention = AttentionFusion_Late(len(self.modality), num_classes=num_classes)
            self.modality_heads = nn.ModuleDict({m: nn.Linear(128, num_classes) for m in self.modality})
        else:
            raise ValueError(f"Unknown fusion_level: {fusion_level}")
        self.dropout_clock = _DropoutClock()

    def forward(self, image=None, radiomics=None, age=None, sex=None, loc=None, artifacts=None):
        clk, tr = self.dropout_clock, self.training
        feats = []
        if 'image' in self.modality:
            feats.append(_run_mlp_ln(self.image_proj, self.image_model(image), clk, 0, tr))
        if 'radiomics' in self.modality:
            feats.append(_run_mlp_ln(self.radiomics_mlp, radiomics, clk, 2, tr))
        if 'clinical' in self.modality:
            # tiny embedding gathers / concat stay in torch (index plumbing, `model.py:186-190`)
            clin = torch.cat([age.unsqueeze(1), self.sex_emb(sex), self.loc_emb(loc)], dim=1)
            feats.append(_run_mlp_ln(self.clinical_mlp, clin, clk, 4, tr))
        if 'artifacts' in self.modality:
            art = torch.cat([self.artifact_embeddings[i](artifacts[:, i]) for i in range(artifacts.size(1))], dim=1)
            feats.append(_run_mlp_ln(self.artifact_mlp, art, clk, 6, tr))
        if tr:
            clk.step += 1
        if self.fusion_level == 'intermediate':
            if self.fusion_strategy == 'concat':
                fused = torch.cat(feats, dim=1)
            elif self.fusion_strategy == 'weighted':
                nw = ops.softmax_rows(self.weights.unsqueeze(0))[0]
                fused = torch.cat([w * f for w, f in zip(nw, feats)], dim=1)
            else:
                fused = self.attention(feats)
            return _run_fusion_mlp(self.fusion_mlp, fused, clk, 8, tr)
        logits = [ops.linear(f, self.modality_heads[m].weight, self.modality_heads[m].bias)
                  for m, f in zip(self.modality, feats)]
        if self.fusion_strategy == 'concat':
            return torch.stack(logits, dim=1).sum(dim=1)
        if self.fusion_strategy == 'weighted':
            nw = ops.softmax_rows(self.weights.unsqueeze(0))[0]
            return torch.stack([w * z for w, z in zip(nw, logits)], dim=0).sum(dim=0)
        return self.attention(logits)


class MultiModalMILNet(nn.Module):
    """Bags of image patches + a radiomic vector per bag -> class logits.

    ``forward(image, radiomics, offsets=None)``: ``image[B, K, 3, H, W]`` (fixed K) or
    ``image[T, 3, H, W]`` with ``offsets[B+1]`` (ragged bags); ``radiomics[B, R]``.
    Returns the teacher head's reference outputs (bag_logits, bag_probs, attention,
    patch_logits, patch_probs; `utils_g_mil.py:99-105`) plus the fused ``logits[B, C]``.
    """

    def __init__(self, hidden_dim=128, att_dim=64, dropout=0.5, radiomics_dim=128, num_classes=7,
                 fusion_strategy='concat', encoder_layers=LAYERS, aux_weight=1.0):
        super().__init__()
        if fusion_strategy not in ('concat', 'attention'):
            raise ValueError(f"Unknown fusion_strategy: {fusion_strategy}")
        self.fusion_strategy, self.aux_weight = fusion_strategy, float(aux_weight)
        self.encoder = ResNet18Encoder(layers=encoder_layers)
        self.mil = AttentionMIL_teacher(self.encoder.out_dim, hidden_dim, att_dim, dropout, num_classes)
        self.image_proj = _mlp_ln(hidden_dim, 256, 128, 0.3, 0.2)
        self.radiomics_mlp = _mlp_ln(radiomics_dim, 256, 128, 0.4, 0.3)
        if fusion_strategy == 'attention':
            self.attention = AttentionFusion(128, 2)
        self.fusion_mlp = nn.Sequential(nn.Linear(256 if fusion_strategy == 'concat' else 128, 256), nn.ReLU(),
                                        nn.Dropout(0.4), nn.Linear(256, num_classes))
        self.dropout_clock = _DropoutClock()

    def set_dropout_state(self, seed, step=0):
        self.dropout_clock.seed, self.dropout_clock.step = int(seed), int(step)
        self.mil.set_dropout_state(seed + 1, step)

    def forward(self, image, radiomics, offsets=None):
        if image.dim() == 5:
            B, K = image.shape[:2]
            image = image.reshape(B * K, *image.shape[2:])
            offs = BagOffsets.uniform(B, K, image.device)
        else:
            offs = as_offsets(offsets, image.device)
        clk, tr = self.dropout_clock, self.training
        feats = self.encoder(image)                                   # [T, 512] fp32
        out = self.mil(feats, offs, return_pooled=True)
        img = _run_mlp_ln(self.image_proj, out.pop("pooled"), clk, 0, tr)
        rad = _run_mlp_ln(self.radiomics_mlp, radiomics, clk, 2, tr)
        fused = torch.cat([img, rad], dim=1) if self.fusion_strategy == 'concat' else self.attention([img, rad])
        out.pop("hidden")
        out["logits"] = _run_fusion_mlp(self.fusion_mlp, fused, clk, 8, tr)
        if tr:
            clk.step += 1
        return out

    def loss(self, out, target):
        """Mean over the step's bags of CE(fused logits) (+ aux_weight * CE(teacher bag logits),
        the per-bag loss of `01_train_mil_teacher.py:244`)."""
        l = ops.cross_entropy(out["logits"], target)
        if self.aux_weight:
            l = l + self.aux_weight * ops.cross_entropy(out["bag_logits"], target)
        return l
