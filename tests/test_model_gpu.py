"""GPU parity of the composed models: MultiModalMILNet vs oracle/model.py, and the
radiomic-fusion net vs the reference-generated golden vectors."""
import numpy as np
import pytest
import torch

from helpers import assert_close, formula_params, load_golden
from oracle import formula, model as omodel, resnet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("R", [32, 128])
@pytest.mark.parametrize("strat", ["concat", "weighted", "attention"])
def test_fusion_net_golden(R, strat):
    """radiomics_mlp / clinical / artifact MLPs + fusion head vs golden vectors produced by the
    reference's own MultiModalFusionNet (model.py:166-227).  fp32 tolerance 5e-5."""
    from model import MultiModalFusionNet
    g = load_golden(f"fusion_{strat}_R{R}.npz")
    net = MultiModalFusionNet(modality=["radiomics", "clinical", "artifacts"], fusion_level="intermediate",
                              fusion_strategy=strat, radiomics_dim=R)
    p = formula_params(g)
    missing = net.load_state_dict({k: v for k, v in p.items() if not k.startswith(("image_model", "image_proj"))},
                                  strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    net = net.to(DEV).eval()
    B = int(g["B"])
    rad = formula.formula_input(B, R, phase=0.9).to(DEV)
    age = formula.ftensor((B,), 0.5, 0.3, 0.1).to(DEV)
    sex, loc = (torch.arange(B) % 3).to(DEV), (torch.arange(B) % 15).to(DEV)
    art = (torch.arange(B * 6).view(B, 6) % 2).to(DEV)
    logits = net(None, rad, age, sex, loc, art)
    assert_close(logits, g["logits"], rtol=5e-5, atol=5e-6, what="logits")


def test_milnet_forward_and_step_vs_oracle():
    """Small ResNet (2 stages) + MIL head + radiomic fusion: forward within bf16 tolerance of the
    oracle (5 % of scale), every head gradient within 10 %, and 3 AdamW steps reduce the loss."""
    from isic_hip import optim
    from model import MultiModalMILNet
    layers = ((64, 1), (128, 2))
    torch.manual_seed(3)
    B, K, S, R, C = 6, 5, 48, 32, 7
    net = MultiModalMILNet(hidden_dim=32, att_dim=16, dropout=0.0, radiomics_dim=R, num_classes=C,
                           encoder_layers=layers).to(DEV)
    net.train()
    net.set_dropout_state(seed=99, step=0)
    g = torch.Generator().manual_seed(5)
    lens = [5, 3, 7, 5, 6, 4]
    offs = np.concatenate([[0], np.cumsum(lens)])
    y = torch.arange(B) % C
    img = torch.randn(int(offs[-1]), 3, S, S, generator=g).bfloat16().float()
    rad = torch.randn(B, R, generator=g)
    p = {k: v.detach().float().cpu().contiguous() for k, v in net.state_dict().items()
         if v.dtype.is_floating_point and "running_" not in k}
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = omodel.milnet_forward(q, img, rad, offs, emulate_bf16=True, layers=layers, drop={"seed": 99, "step": 0})
    lref = omodel.milnet_loss(ref, y)
    lref.backward()
    out = net(img.to(DEV), rad.to(DEV), offsets=offs)
    loss = net.loss(out, y.to(DEV))
    loss.backward()
    for k in ("bag_logits", "logits", "attention"):
        r = ref[k].detach()
        err = float((out[k].detach().cpu() - r).abs().max())
        assert err < 0.05 * float(r.abs().max()) + 1e-3, (k, err)
    assert abs(float(loss.detach()) - float(lref)) < 0.03
    for k, prm in net.named_parameters():
        if k.startswith("encoder") or k == "mil.attention.2.bias":
            continue   # encoder: tests/test_encoder_gpu.py; attention.2.bias: analytically zero gradient (softmax shift invariance)
        gr = q[k].grad
        rel = float((prm.grad.cpu() - gr).norm() / (gr.norm() + 1e-9))
        assert rel < 0.10, (k, rel)
    opt = optim.AdamW(net.parameters(), lr=2e-3, weight_decay=1e-4)
    first = last = None
    for it in range(4):
        opt.zero_grad()
        out = net(img.to(DEV), rad.to(DEV), offsets=offs)
        l = net.loss(out, y.to(DEV))
        l.backward()
        opt.step()
        last = float(l.detach())
        first = last if first is None else first
    assert last < first
