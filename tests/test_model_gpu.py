"""GPU parity of the composed models: MultiModalMILNet vs oracle/model.py, and the
radiomic-fusion net vs the reference-generated golden vectors."""
import numpy as np
import pytest
import torch

from helpers import assert_close, formula_params, load_golden
from oracle import formula, model as omodel, resnet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("level", ["intermediate", "late"])
@pytest.mark.parametrize("R", [32, 128])
@pytest.mark.parametrize("strat", ["concat", "weighted", "attention"])
def test_fusion_net_golden(R, strat, level):
    """radiomics_mlp / clinical / artifact MLPs + the fusion branches of BOTH levels (intermediate: `model.py:206-216`;
    late: per-modality heads `:155-164`, sum / softmax-weighted / AttentionFusion_Late `:25-40`, `:216-227`) vs golden
    vectors produced by the reference's own MultiModalFusionNet: logits (fp32 tolerance 5e-5), the CE loss, and the
    gradient of every parameter the reference's backward reached (3e-4 of the tensor's scale)."""
    from helpers import check_grad
    from model import MultiModalFusionNet
    g = load_golden(f"fusion_{strat}_R{R}.npz" if level == "intermediate" else f"fusion_late_{strat}_R{R}.npz")
    net = MultiModalFusionNet(modality=["radiomics", "clinical", "artifacts"], fusion_level=level,
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
    loss = torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(g["target"]).to(DEV))   # as net_utils.train does
    assert abs(float(loss.detach()) - float(g["loss"])) < 5e-5
    loss.backward()
    keys = sorted({k[5:].split("#")[0] for k in g.files if k.startswith("grad.")})
    params = dict(net.named_parameters())
    assert keys and all(k in params for k in keys)
    for k in keys:
        gr = params[k].grad
        assert gr is not None, k
        if k == "attention.attn.2.bias":
            # analytically zero (the softmax over the modalities is invariant to a shift of every score): the reference itself
            # holds 4.5e-8 of rounding noise here, and so does this path -- compared absolutely
            assert float(gr.abs().max()) < 1e-6 and abs(float(g[f"grad.{k}"].reshape(-1)[0])) < 1e-6
            continue
        check_grad(g, k, gr, rtol=3e-4, atol=1e-7)
    for k, prm in params.items():
        if k not in keys:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, k
    if strat == "attention" and level == "late":
        lg = [formula.formula_input(B, 7, phase=1.7 + 1.2 * i).to(DEV) for i in range(3)]
        assert_close(net.attention(lg), g["attfusion_late"], rtol=5e-5, atol=5e-6, what="attfusion_late")


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


def test_net_utils_loops_match_reference_golden(capsys):
    """`net_utils.train / validate / test` (reference `net_utils.py:6-127`) driving MultiModalFusionNet on the HIP
    path == the reference's own loops driving the reference's own model on CPU (tests/golden/net_utils_loops.npz,
    generated by oracle/gen_golden.py): parameters after two SGD epochs, the returned validation loss, the test
    accuracy, the classification report and the printed epoch lines."""
    import net_utils as nu
    from helpers import shapes_from_blob
    from model import MultiModalFusionNet
    g = load_golden("net_utils_loops.npz")
    R, B = int(g["R"]), int(g["B"])
    net = MultiModalFusionNet(modality=["radiomics", "clinical", "artifacts"], fusion_level="intermediate",
                              fusion_strategy="concat", radiomics_dim=R)
    p = formula_params(g)
    missing = net.load_state_dict({k: v for k, v in p.items() if not k.startswith(("image_model", "image_proj"))},
                                  strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    for mod in net.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    net = net.to(DEV)

    def make_batch(i):       # the generator's closed-form batches (oracle/gen_golden.py)
        return {"image": torch.zeros(B, 1), "radiomics": formula.formula_input(B, R, phase=0.3 + 0.5 * i),
                "age": formula.ftensor((B,), 0.5, 0.3, 0.1 + i), "sex": (torch.arange(B) + i) % 3,
                "loc": (torch.arange(B) * 2 + i) % 15, "artifacts": ((torch.arange(B * 6).view(B, 6) + i) % 2),
                "target": (torch.arange(B) * 3 + i) % 7}

    loader, held = [make_batch(i) for i in range(3)], [make_batch(5)]
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    for ep in range(2):
        nu.train(net, loader, crit, opt, DEV, None, ep)
    val = nu.validate(net, held, crit, DEV, None, 2)
    acc, report = nu.test(net, held, DEV, None)
    assert abs(val - float(g["val_loss"])) < 2e-4 * max(1.0, abs(float(g["val_loss"])))
    assert acc == float(g["test_acc"])
    assert report == str(g["report"])
    out = capsys.readouterr().out
    assert out.strip().splitlines() == str(g["stdout"]).strip().splitlines()
    for k, v in net.state_dict().items():
        if f"after.{k}" in g.files:
            assert_close(v, g[f"after.{k}"], rtol=2e-4, atol=2e-6, what=k)
        elif f"after.{k}#sample" in g.files:
            a = v.detach().cpu().double().numpy().reshape(-1)
            stride = int(g[f"after.{k}#stride"])
            ref = g[f"after.{k}#sample"].astype(np.float64)
            assert_close(a[::stride][: ref.size], ref, rtol=2e-4, atol=2e-6, what=k)
            assert a.size == int(g[f"after.{k}#numel"])
