"""Encoder parity where round 1 was loose: (a) every BasicBlock's backward, teacher-forced -- the block is fed an
exact input and upstream gradient and compared with torch autograd of the same fp32 block (<= 2 % per tensor),
so a systematic error in one block cannot hide behind the chaos of 17 stacked bf16 layers; (b) BASELINE.json
configs[1] AT ITS REAL SIZE (1024 and 2048 images of 3x224x224 through the whole encoder, forward and backward):
fused BatchNorm statistics == statistics of the written tensor for every layer, sampled output pixels of every
convolution vs an fp32 recomputation from the layer's own input, finite gradients, run-to-run equality of the
deterministic kernels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


# (block prefix, Cin, planes, stride, input H = W, images): every block of ResNet-18, the three stride-2 /
# downsample blocks included; >= 256 samples per channel behind every BatchNorm
BLOCKS = [("layer1.0", 64, 64, 1, 16, 4), ("layer1.1", 64, 64, 1, 12, 6), ("layer2.0", 64, 128, 2, 16, 8),
          ("layer2.1", 128, 128, 1, 10, 6), ("layer3.0", 128, 256, 2, 12, 12), ("layer3.1", 256, 256, 1, 7, 8),
          ("layer4.0", 256, 512, 2, 8, 24), ("layer4.1", 512, 512, 1, 7, 8)]


def _rb(t):
    return t.bfloat16().float()


def _bn_stats(c, gamma, beta, eps=1e-5):
    """Training-mode statistics of the (already rounded) conv output, as `isic_bn_finalize` forms them."""
    mean = c.mean(dim=(0, 2, 3), dtype=torch.float64)
    var = (c.double() - mean.view(1, -1, 1, 1)).pow(2).mean(dim=(0, 2, 3))
    rstd = (var + eps).rsqrt()
    scale = (gamma.double() * rstd).float()
    shift = (beta.double() - mean * gamma.double() * rstd).float()
    return mean.float(), rstd.float(), scale, shift


def _bn_bwd(dz, c, mean, rstd, gamma):
    """(dc rounded to bf16, dgamma, dbeta) of y = gamma * xhat + beta given dz = d loss / d y."""
    v = lambda t: t.view(1, -1, 1, 1)
    xh = (c - v(mean)) * v(rstd)
    R = dz.numel() // dz.shape[1]
    dbeta = dz.sum(dim=(0, 2, 3), dtype=torch.float64)
    dgamma = (dz * xh).sum(dim=(0, 2, 3), dtype=torch.float64)
    dc = v(gamma * rstd) * (dz - v((dbeta / R).float()) - xh * v((dgamma / R).float()))
    return _rb(dc), dgamma.float(), dbeta.float()


@pytest.mark.parametrize("blk", BLOCKS, ids=[b[0] for b in BLOCKS])
def test_block_backward_teacher_forced(blk):
    """HIP BasicBlock forward + backward (`ResNet18Encoder.block_forward / block_backward`) on an exact input, exact
    weights and an exact upstream gradient vs a torch-CPU fp32 restatement of the SAME dataflow -- every tensor the HIP
    path stores as bf16 (c1, a1, c2, the block output, dc2, da1, dc1, the identity / downsample gradients) is rounded
    at the same point, convolutions and their gradients come from torch (F.conv2d, torch.nn.grad).  What is left is
    fp32 summation order and values that straddle a rounding boundary: <= 2 % per tensor (norm-wise) for the input
    gradient and every parameter gradient, <= 1 % for the block output."""
    from torch.nn.grad import conv2d_input, conv2d_weight
    from isic_hip.encoder import ResNet18Encoder
    pre, cin, planes, stride, H, N = blk
    ds = stride != 1 or cin != planes
    torch.manual_seed(sum(map(ord, pre)))
    enc = ResNet18Encoder().to(DEV)
    enc.train()
    g = torch.Generator().manual_seed(7)
    names = [f"{pre}.conv1.weight", f"{pre}.bn1.weight", f"{pre}.bn1.bias", f"{pre}.conv2.weight", f"{pre}.bn2.weight",
             f"{pre}.bn2.bias"] + ([f"{pre}.downsample.0.weight", f"{pre}.downsample.1.weight",
                                    f"{pre}.downsample.1.bias"] if ds else [])
    P = {}
    with torch.no_grad():
        for k in names:
            p = enc._get(k)
            if p.dim() == 4:       # the kernels see bf16 weights: make them exact
                v = _rb(torch.randn(p.shape, generator=g) * float(np.sqrt(2.0 / (p.shape[0] * p.shape[2] * p.shape[3]))))
            elif k.endswith("weight"):
                v = 1.0 + 0.2 * torch.randn(p.shape, generator=g)
            else:
                v = 0.1 * torch.randn(p.shape, generator=g)
            p.copy_(v.to(DEV).contiguous(memory_format=torch.channels_last) if p.dim() == 4 else v.to(DEV))
            P[k[len(pre) + 1:]] = v.clone()
    x = _rb(torch.relu(torch.randn(N, cin, H, H, generator=g)))      # a post-ReLU activation
    Ho = (H + 2 - 3) // stride + 1
    gout = _rb(1.0 + 0.5 * torch.randn(N, planes, Ho, Ho, generator=g))
    # ---- torch fp32 restatement with the HIP path's bf16 roundings
    w1, w2 = P["conv1.weight"], P["conv2.weight"]
    c1 = _rb(F.conv2d(x, w1, None, stride, 1))
    m1, r1, sc1, sh1 = _bn_stats(c1, P["bn1.weight"], P["bn1.bias"])
    v = lambda t: t.view(1, -1, 1, 1)
    pre1 = c1 * v(sc1) + v(sh1)
    a1 = _rb(torch.relu(pre1))
    c2 = _rb(F.conv2d(a1, w2, None, 1, 1))
    m2, r2, sc2, sh2 = _bn_stats(c2, P["bn2.weight"], P["bn2.bias"])
    idn = x
    if ds:
        wd = P["downsample.0.weight"]
        cd = _rb(F.conv2d(x, wd, None, stride, 0))
        md, rd, scd, shd = _bn_stats(cd, P["downsample.1.weight"], P["downsample.1.bias"])
        idn = _rb(cd * v(scd) + v(shd))
    out_ref = _rb(torch.relu(c2 * v(sc2) + v(sh2) + idn))
    G = {}
    dz2 = gout * (out_ref > 0).float()
    dc2, G["bn2.weight"], G["bn2.bias"] = _bn_bwd(dz2, c2, m2, r2, P["bn2.weight"])
    dres = dz2                                                        # bf16-exact: a masked copy of the upstream gradient
    G["conv2.weight"] = conv2d_weight(a1, w2.shape, dc2, 1, 1)
    da1 = _rb(conv2d_input(a1.shape, w2, dc2, 1, 1))
    dz1 = da1 * (pre1 > 0).float()                                    # the mask is recomputed from c1*scale+shift
    dc1, G["bn1.weight"], G["bn1.bias"] = _bn_bwd(dz1, c1, m1, r1, P["bn1.weight"])
    G["conv1.weight"] = conv2d_weight(x, w1.shape, dc1, stride, 1)
    dx_main = conv2d_input(x.shape, w1, dc1, stride, 1)
    if ds:
        dcd, G["downsample.1.weight"], G["downsample.1.bias"] = _bn_bwd(dres, cd, md, rd, P["downsample.1.weight"])
        G["downsample.0.weight"] = conv2d_weight(x, wd.shape, dcd, stride, 0)
        dx_ref = _rb(dx_main + _rb(conv2d_input(x.shape, wd, dcd, stride, 0)))     # joined in fp32, one rounding
    else:
        dx_ref = _rb(dx_main + dres)
    # ---- HIP block
    enc.prepare_weights()
    enc._arena_reset(torch.device(DEV))
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV, BF)
    out, saved = enc.block_forward(xd, pre, ds)
    assert _rel(out.float().cpu().permute(0, 3, 1, 2), out_ref) < 0.01
    for k in names:
        enc._get(k).grad = None
    enc._arena_reset(torch.device(DEV))
    dx, done = enc.block_backward(gout.permute(0, 2, 3, 1).contiguous().to(DEV, BF), pre, ds, saved)
    torch.cuda.synchronize()
    assert set(done) == set(names)
    errs = {"dx": _rel(dx.float().cpu().permute(0, 3, 1, 2), dx_ref)}
    for k in names:
        errs[k] = _rel(enc._get(k).grad.cpu(), G[k[len(pre) + 1:]])
    bad = {k: round(e, 4) for k, e in errs.items() if e > 0.02}
    assert not bad, f"{pre}: relative gradient errors above 2 %: {bad} (all: { {k: round(e, 4) for k, e in errs.items()} })"


def _conv_samples(x, wf, out, spec, n_samples, gen):
    """fp32 recomputation of `n_samples` random output pixels (all output channels) of one convolution from the
    layer's own bf16 input and bf16 weights; returns max |diff| / (2^-7 |ref| + small)."""
    N, H, W, C = x.shape
    _, Ho, Wo, Co = out.shape
    k, s, p = spec.k, spec.stride, spec.pad
    n = torch.randint(0, N, (n_samples,), generator=gen, device=DEV)
    ho = torch.randint(0, Ho, (n_samples,), generator=gen, device=DEV)
    wo = torch.randint(0, Wo, (n_samples,), generator=gen, device=DEV)
    w = wf.view(Co, k, k, C).float()
    acc = torch.zeros(n_samples, Co, device=DEV, dtype=torch.float32)
    for kh in range(k):
        for kw in range(k):
            hi, wi = ho * s + kh - p, wo * s + kw - p
            ok = (hi >= 0) & (hi < H) & (wi >= 0) & (wi < W)
            px = x[n, hi.clamp(0, H - 1), wi.clamp(0, W - 1)].float() * ok.view(-1, 1).float()      # [S, C]
            acc += px @ w[:, kh, kw, :].t()
    got = out[n, ho, wo].float()
    tol = (2.0 ** -7) * acc.abs() + 2e-3 * float(acc.abs().max()) * 2 ** -4
    return float(((got - acc).abs() / tol).max())


@pytest.mark.parametrize("images", [1024, 2048])
def test_configs1_full_size_forward_backward(images):
    """16 and 32 bags x 64 patches of 3x224x224 (bench.py's per-GPU step) through the whole ResNet-18."""
    from isic_hip.encoder import ResNet18Encoder
    torch.manual_seed(5)
    enc = ResNet18Encoder().to(DEV)
    enc.train()
    gen = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(images, 3, 224, 224, device=DEV, generator=gen).to(BF)
    feat, tape = enc.run_forward(x, save=True)
    assert feat.shape == (images, 512) and bool(torch.isfinite(feat).all())
    # ---- every convolution: sampled outputs vs fp32 recomputation; every BatchNorm: fused statistics == statistics
    #      of the tensor that was written (mean / rstd are what the backward passes and the apply kernels use)
    worst = {}

    def check_bn(c, st, name):
        mean, rstd = st[0], st[1]
        cf = c.float().view(-1, c.shape[-1])
        m = cf.mean(0, dtype=torch.float64)
        v = (cf.double() - m).pow(2).mean(0)
        assert float((mean.double() - m).abs().max()) <= 1e-5 * float(m.abs().max()) + 1e-6, name
        r = (v + 1e-5).rsqrt()
        assert float(((rstd.double() - r).abs() / r).max()) <= 1e-4, name

    c, st0, am, _, _ = tape["stem"]
    x0 = tape["x0"]                                      # NHWC4 bf16
    ws, _ = enc._weights("conv1", False)
    check_bn(c, st0, "bn1")
    for (pre, ds), saved in zip(enc.blocks, tape["blocks"]):
        xin, c1, a1, st1, c2, out, st2, cd, std = saved
        for nm, inp, o in ((f"{pre}.conv1", xin, c1), (f"{pre}.conv2", a1, c2)) + (((f"{pre}.downsample.0", xin, cd),) if ds else ()):
            wf, _ = enc._weights(nm, False)
            worst[nm] = _conv_samples(inp, wf, o, enc.specs[nm], 256, gen)
        check_bn(c1, st1, f"{pre}.bn1")
        check_bn(c2, st2, f"{pre}.bn2")
        if ds:
            check_bn(cd, std, f"{pre}.downsample.1")
        assert bool(torch.isfinite(out.float()).all()), pre
    bad = {k: round(v, 3) for k, v in worst.items() if v > 1.0}
    assert not bad, f"sampled conv outputs beyond one bf16 rounding of the fp32 recomputation: {bad}"
    # ---- backward: finite gradients everywhere, and run-to-run equality where the kernels are deterministic
    dfeat = torch.randn(images, 512, device=DEV, generator=gen) / images

    def grads():
        for p in enc.parameters():
            p.grad = None
        enc.run_backward(tape, dfeat)
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in enc.named_parameters()}
    g1 = grads()
    g2 = grads()
    for k, v in g1.items():
        assert bool(torch.isfinite(v).all()) and float(v.abs().max()) > 0, k
    for k in g1:
        a, b = g1[k], g2[k]
        if k.startswith(("layer1.", "layer2.1", "layer2.0.conv2", "layer3.1", "layer3.0.conv2", "layer4.1", "layer4.0.conv2")) and \
                k.endswith(("conv1.weight", "conv2.weight")):          # every 3x3 / stride-1 layer: all-taps kernels, no atomics
            assert torch.equal(a, b), f"{k}: the all-taps weight gradient must be deterministic"
        else:
            assert _rel(a, b) < 1e-4, k               # fp32 / fp64 atomics: last-bit differences only


@pytest.mark.parametrize("images", [96, 2048])
def test_training_step_is_bit_reproducible(images):
    """the SAME step run twice -- forward from the images, backward from the same feature gradient --
    must give bit-identical features, BatchNorm statistics and EVERY encoder parameter gradient, at the configs[1] size
    (2048 images of 224x224: every persistent kernel at its full grid) and at a small size (few tiles per block, the
    stem below its block cap).  What made two runs differ before round 3: LDS / fp64 atomics in arrival order inside the
    fused BatchNorm statistics of conv_halo / conv3x3_c64p / conv_pgemm / the stem, and fp32 atomics in 7 of the 20
    weight gradients; 17 bf16 layers amplified those last-bit differences to per cent in the gradients."""
    from isic_hip.encoder import ResNet18Encoder
    torch.manual_seed(5)
    enc = ResNet18Encoder().to(DEV)
    enc.train()
    gen = torch.Generator(device=DEV).manual_seed(13)
    S = 224 if images >= 1024 else 96
    x = torch.randn(images, 3, S, S, device=DEV, generator=gen).to(BF)
    dfeat = torch.randn(images, 512, device=DEV, generator=gen) / images

    def run():
        for p in enc.parameters():
            p.grad = None
        feat, tape = enc.run_forward(x, save=True)
        stats = [tape["stem"][1][0].clone(), tape["stem"][1][1].clone()]
        for saved in tape["blocks"]:
            for st in (saved[3], saved[6], saved[8]):
                if st is not None:
                    stats += [st[0].clone(), st[1].clone()]
        enc.run_backward(tape, dfeat)
        torch.cuda.synchronize()
        return feat.clone(), stats, {k: p.grad.detach().clone() for k, p in enc.named_parameters()}

    f1, s1, g1 = run()
    f2, s2, g2 = run()
    assert torch.equal(f1, f2), "features differ between two identical forwards"
    assert all(torch.equal(a, b) for a, b in zip(s1, s2)), "BatchNorm batch statistics differ between two identical forwards"
    diff = [k for k in g1 if not torch.equal(g1[k], g2[k])]
    assert not diff, f"parameter gradients differ between two identical steps: {diff}"
    assert all(bool(torch.isfinite(v).all()) and float(v.abs().max()) > 0 for v in g1.values())
    # round 4: the gradient through an identity block's skip joined from (d out, ReLU mask) inside conv1's data gradient
    # (isic_conv2d_igemm_maskadd_bf16) == written by bn2's backward and read back as an addend: a masked bf16 value is the
    # value or zero, so EVERY gradient is bit-identical
    assert enc.mask_identity_gradient
    enc.mask_identity_gradient = False
    f3, s3, g3 = run()
    enc.mask_identity_gradient = True
    diff = [k for k in g1 if not torch.equal(g1[k], g3[k])]
    assert not diff, f"masked identity-gradient join changes the gradients: {diff}"
    # ... and the downsample shortcut's BatchNorm applied inside bn2's apply pass (isic_bn_apply_mask_res_affine_bf16: the inner
    # value is rounded to bf16 as the pass that materialised it did) == the materialised shortcut: features, statistics, gradients
    assert enc.fold_shortcut_norm
    enc.fold_shortcut_norm = False
    f4, s4, g4 = run()
    enc.fold_shortcut_norm = True
    assert torch.equal(f1, f4) and all(torch.equal(a, b) for a, b in zip(s1, s4))
    diff = [k for k in g1 if not torch.equal(g1[k], g4[k])]
    assert not diff, f"folding the shortcut's BatchNorm changes the gradients: {diff}"


def _ref_block(x, P, stride, ds, forced=None):
    """torch-CPU fp32 restatement of one BasicBlock with the HIP path's bf16 rounding points (the dataflow of
    test_block_backward_teacher_forced, as a function): returns (out, backward) with backward(gout) -> (dx, grads).
    ``forced`` = the HIP block's own saved forward tensors (c1, a1, c2, out, cd as NCHW fp32): the forward is then TAKEN
    from the HIP path (teacher forcing) and only the backward is restated -- the comparison is not blurred by the
    forward's rounding flips."""
    from torch.nn.grad import conv2d_input, conv2d_weight
    v = lambda t: t.view(1, -1, 1, 1)
    w1, w2 = P["conv1.weight"], P["conv2.weight"]
    c1 = forced["c1"] if forced else _rb(F.conv2d(x, w1, None, stride, 1))
    m1, r1, sc1, sh1 = _bn_stats(c1, P["bn1.weight"], P["bn1.bias"])
    pre1 = c1 * v(sc1) + v(sh1)
    a1 = forced["a1"] if forced else _rb(torch.relu(pre1))
    c2 = forced["c2"] if forced else _rb(F.conv2d(a1, w2, None, 1, 1))
    m2, r2, sc2, sh2 = _bn_stats(c2, P["bn2.weight"], P["bn2.bias"])
    idn = x
    if ds:
        wd = P["downsample.0.weight"]
        cd = forced["cd"] if forced else _rb(F.conv2d(x, wd, None, stride, 0))
        md, rd, scd, shd = _bn_stats(cd, P["downsample.1.weight"], P["downsample.1.bias"])
        idn = _rb(cd * v(scd) + v(shd))
    out = forced["out"] if forced else _rb(torch.relu(c2 * v(sc2) + v(sh2) + idn))

    def backward(gout):
        G = {}
        dz2 = gout * (out > 0).float()
        dc2, G["bn2.weight"], G["bn2.bias"] = _bn_bwd(dz2, c2, m2, r2, P["bn2.weight"])
        G["conv2.weight"] = conv2d_weight(a1, w2.shape, dc2, 1, 1)
        da1 = _rb(conv2d_input(a1.shape, w2, dc2, 1, 1))
        dz1 = da1 * (pre1 > 0).float()
        dc1, G["bn1.weight"], G["bn1.bias"] = _bn_bwd(dz1, c1, m1, r1, P["bn1.weight"])
        G["conv1.weight"] = conv2d_weight(x, w1.shape, dc1, stride, 1)
        dx_main = conv2d_input(x.shape, w1, dc1, stride, 1)
        if ds:
            dcd, G["downsample.1.weight"], G["downsample.1.bias"] = _bn_bwd(dz2, cd, md, rd, P["downsample.1.weight"])
            G["downsample.0.weight"] = conv2d_weight(x, wd.shape, dcd, stride, 0)
            dx = _rb(dx_main + _rb(conv2d_input(x.shape, wd, dcd, stride, 0)))
        else:
            dx = _rb(dx_main + dz2)
        return dx, G
    return out, backward


def test_three_block_chain_backward():
    """the COMPOSITION of blocks -- layer1.1 -> layer2.0 (stride 2 + 1x1 downsample) -> layer2.1 run as
    a chain on the HIP path, forward and backward, against a chain of torch-CPU fp32 blocks that round to bf16 at the same
    points.  Two references:
      * free-running: the reference chain computes its own forward.  Its activations differ from the HIP chain's by
        rounding flips (<= 1 % of the output after three blocks), which the ReLU masks and BatchNorm statistics of the
        backward amplify: every gradient within 8 % (measured 5-6 %);
      * forward teacher-forced, backward free-running: every reference block takes its forward tensors from the HIP
        chain, but the BACKWARD runs as a chain -- block i's dx is block i-1's upstream gradient, the residual-gradient
        joins and the 64 -> 128 channel / stride-2 hand-over are inside the comparison and nothing is re-injected between
        blocks: <= 3 % (norm-wise) for the chain's input gradient and every parameter gradient of the three blocks."""
    from isic_hip.encoder import ResNet18Encoder
    chain = [("layer1.1", 64, 64, 1), ("layer2.0", 64, 128, 2), ("layer2.1", 128, 128, 1)]
    N, H = 12, 24
    torch.manual_seed(77)
    enc = ResNet18Encoder().to(DEV)
    enc.train()
    g = torch.Generator().manual_seed(17)
    Ps, names_all = [], []
    with torch.no_grad():
        for pre, cin, planes, stride in chain:
            ds = stride != 1 or cin != planes
            names = [f"{pre}.conv1.weight", f"{pre}.bn1.weight", f"{pre}.bn1.bias", f"{pre}.conv2.weight", f"{pre}.bn2.weight",
                     f"{pre}.bn2.bias"] + ([f"{pre}.downsample.0.weight", f"{pre}.downsample.1.weight",
                                            f"{pre}.downsample.1.bias"] if ds else [])
            P = {}
            for k in names:
                p = enc._get(k)
                if p.dim() == 4:
                    val = _rb(torch.randn(p.shape, generator=g) * float(np.sqrt(2.0 / (p.shape[0] * p.shape[2] * p.shape[3]))))
                elif k.endswith("weight"):
                    val = 1.0 + 0.2 * torch.randn(p.shape, generator=g)
                else:
                    val = 0.1 * torch.randn(p.shape, generator=g)
                p.copy_(val.to(DEV).contiguous(memory_format=torch.channels_last) if p.dim() == 4 else val.to(DEV))
                P[k[len(pre) + 1:]] = val.clone()
            Ps.append(P)
            names_all.append(names)
    x = _rb(torch.relu(torch.randn(N, 64, H, H, generator=g)))
    # ---- HIP chain
    enc.prepare_weights()
    enc._arena_reset(torch.device(DEV))
    h, saved = x.permute(0, 2, 3, 1).contiguous().to(DEV, BF), []
    for pre, cin, planes, stride in chain:
        h, s = enc.block_forward(h, pre, stride != 1 or cin != planes)
        saved.append(s)
    out_hip = h.float().cpu().permute(0, 3, 1, 2)
    gout = _rb(1.0 + 0.5 * torch.randn(out_hip.shape, generator=g))
    for names in names_all:
        for k in names:
            enc._get(k).grad = None
    enc._arena_reset(torch.device(DEV))
    gd = gout.permute(0, 2, 3, 1).contiguous().to(DEV, BF)
    for (pre, cin, planes, stride), s in zip(reversed(chain), reversed(saved)):
        gd, _ = enc.block_backward(gd, pre, stride != 1 or cin != planes, s)
    torch.cuda.synchronize()
    dx_hip = gd.float().cpu().permute(0, 3, 1, 2)
    nchw = lambda t: None if t is None else t.float().cpu().permute(0, 3, 1, 2).contiguous()

    def reference(forced_mode):
        t, bwds = x, []
        for (pre, cin, planes, stride), P, sv in zip(chain, Ps, saved):
            forced = None
            if forced_mode:
                xin, c1, a1, _st1, c2, out, _st2, cd, _std = sv
                t = nchw(xin)
                forced = {"c1": nchw(c1), "a1": nchw(a1), "c2": nchw(c2), "out": nchw(out), "cd": nchw(cd)}
            t, b = _ref_block(t, P, stride, stride != 1 or cin != planes, forced)
            bwds.append(b)
        gr, Gs = gout, []
        for b in reversed(bwds):
            gr, G = b(gr)
            Gs.append(G)
        Gs.reverse()
        return t, gr, Gs

    for forced_mode, tol in ((False, 0.08), (True, 0.03)):
        out_ref, dx_ref, Gs = reference(forced_mode)
        assert _rel(out_hip, out_ref) < 0.01
        errs = {"dx": _rel(dx_hip, dx_ref)}
        for (pre, *_), names, G in zip(chain, names_all, Gs):
            for k in names:
                errs[k] = _rel(enc._get(k).grad.cpu(), G[k[len(pre) + 1:]])
        bad = {k: round(e, 4) for k, e in errs.items() if e > tol}
        what = "forward teacher-forced" if forced_mode else "free-running"
        print(f"[3-block chain, {what}] max rel err {max(errs.values()):.4f}")
        assert not bad, f"3-block chain ({what}): relative gradient errors above {tol}: {bad} (all: { {k: round(e, 4) for k, e in errs.items()} })"
