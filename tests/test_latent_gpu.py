"""Latent extraction path (`save_latent.py:13-204`): patch flags and the patch-level frame vs the
line-by-line CPU restatement (oracle/latents.py), the token grid vs the encoder's own training-path activations, and
the drop-in ``extract_latents`` return contract on synthetic ISIC-shaped images."""
import numpy as np
import pytest
import torch

from oracle import latents as olat

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_mask_patch_flags_bit_exact():
    import save_latent as sl
    g = torch.Generator().manual_seed(0)
    mask = (torch.rand(7, 1, 224, 224, generator=g) > 0.9995).float()        # sparse: many empty patches
    mask[2] = 0
    mask[3, 0, 100:140, 30:90] = 0.3                                         # any value > 0 counts
    mask[4, 0, 223, 223] = 1.0                                               # a single pixel in the last patch
    got = sl.mask_patch_flags(mask.to(DEV)).cpu()
    assert got.shape == (7, 14, 14) and got.dtype == torch.bool
    assert torch.equal(got, olat.mask_patch_flags(mask))
    assert torch.equal(sl.mask_patch_flags(mask[:, 0].to(DEV)).cpu(), got)   # (B,H,W) form, save_latent.py:74-75
    m2 = (torch.rand(3, 64, 96, generator=g) > 0.99).float()
    assert torch.equal(sl.mask_patch_flags(m2.to(DEV), patch=8).cpu(), olat.mask_patch_flags(m2, 8))


def test_tokens_are_the_layer3_activations():
    """`run_tokens` (eval-mode, frozen) == the stage-3 block output of the encoder's own forward in eval mode."""
    from isic_hip.encoder import ResNet18Encoder
    torch.manual_seed(1)
    enc = ResNet18Encoder().to(DEV).eval()
    x = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(2)).to(DEV)
    tok = enc.run_tokens(x)
    assert tok.shape == (3, 196, 256) and tok.dtype == torch.float32 and bool(torch.isfinite(tok).all())
    feat, tape = enc.run_forward(x, save=True)                               # eval mode: running statistics as well
    assert torch.equal(tok, tape["blocks"][5][5].float().view(3, 196, 256))
    with pytest.raises(Exception):
        enc.train().run_tokens(x)


def test_extract_latents_contract_and_patch_frames():
    import save_latent as sl
    cfg = {"device": DEV, "seed": 42, "pca": False}
    tv, te = sl.SyntheticDermImages(n=11, seed=1), sl.SyntheticDermImages(n=6, seed=2)
    out = sl.extract_latents(cfg, "missing.pth", remove_background=True, datasets=(tv, te), batch_size=4)
    ptr, pte, pool_tr, pool_te, raw_tr, raw_te = out
    assert list(pool_tr.columns) == ["image_path", "segmentation_path", "target", "latent_pooled_max", "latent_pooled_mean",
                                     "ids_restore", "ids_keep"]                                   # save_latent.py:65-73
    assert list(raw_tr.columns) == ["image_path", "segmentation_path", "target", "latent", "ids_restore", "ids_keep",
                                    "lesion_mask_patches"]                                        # :89-97
    assert list(ptr.columns) == ["image_path", "segmentation_path", "target", "patch_id", "patch_latent", "patch_in_mask",
                                 "patch_latent_pca"]                                              # :135-142,182-185
    assert len(pool_tr) == 11 and len(raw_te) == 6 and raw_tr["latent"].iloc[0].shape == (196, 256)
    lat0 = raw_tr["latent"].iloc[3]
    assert np.allclose(pool_tr["latent_pooled_max"].iloc[3], lat0.max(axis=0)) and \
        np.allclose(pool_tr["latent_pooled_mean"].iloc[3], lat0.mean(axis=0), atol=1e-6)
    # patch flags of every image == the reference's unfold logic on the same masks
    for i in range(11):
        assert np.array_equal(raw_tr["lesion_mask_patches"].iloc[i], olat.mask_patch_flags(tv[i]["mask"][None])[0].numpy())
    assert not raw_tr["lesion_mask_patches"].iloc[4].any()                   # the 'no_mask' image keeps no patch
    # patch-level frames == the reference's per-patch double loop, remove on and off
    for raw, got in ((raw_tr, ptr), (raw_te, pte)):
        ref, cnt = olat.build_patch_level_df(raw, remove=True)
        assert len(ref) == len(got) == cnt and cnt > 0
        for c in ("image_path", "segmentation_path", "target", "patch_id", "patch_in_mask"):
            assert list(ref[c]) == list(got[c]), c
        assert all(np.array_equal(a, b) for a, b in zip(ref["patch_latent"], got["patch_latent"]))
        assert all(a is b or np.array_equal(a, b) for a, b in zip(got["patch_latent_pca"], got["patch_latent"]))
    full, cnt0 = sl.build_patch_level_df(raw_te, remove=False)
    ref_full, rc0 = olat.build_patch_level_df(raw_te, remove=False)
    assert len(full) == 6 * 196 == len(ref_full) and cnt0 == rc0 == 0
    assert list(full["patch_in_mask"]) == list(ref_full["patch_in_mask"]) and list(full["patch_id"]) == list(ref_full["patch_id"])
    # PCA branch (save_latent.py:163-180)
    out2 = sl.extract_latents({"device": DEV, "seed": 42, "pca": True}, "missing.pth", True, datasets=(tv, te), batch_size=8)
    d = out2[0]["patch_latent_pca"].iloc[0].shape[0]
    assert 0 < d < 256 and out2[1]["patch_latent_pca"].iloc[0].shape[0] == d


def test_extract_latents_with_the_vit_s16_encoder():
    """BASELINE.json configs[4]: ``encoder: vit_s16`` in the config swaps the frozen encoder for the fp16 ViT-S/16 --
    same return contract, 196 tokens of 384 channels."""
    import save_latent as sl
    cfg = {"device": DEV, "seed": 42, "pca": False, "encoder": "vit_s16"}
    tv, te = sl.SyntheticDermImages(n=5, seed=1), sl.SyntheticDermImages(n=3, seed=2)
    ptr, pte, pool_tr, pool_te, raw_tr, raw_te = sl.extract_latents(cfg, "missing.pth", datasets=(tv, te), batch_size=4)
    assert len(pool_tr) == 5 and len(raw_te) == 3 and raw_tr["latent"].iloc[0].shape == (196, 384)
    lat = np.stack(list(raw_tr["latent"]))
    assert np.isfinite(lat).all() and lat.std() > 0.1
    assert np.allclose(pool_tr["latent_pooled_mean"].iloc[2], raw_tr["latent"].iloc[2].mean(axis=0), atol=1e-5)
    assert len(ptr) == 5 * 196 and ptr["patch_latent"].iloc[0].shape == (384,)
