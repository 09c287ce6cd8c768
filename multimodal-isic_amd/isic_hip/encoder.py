"""ResNet-18 patch encoder on hand-written HIP kernels (bf16 MFMA, NHWC).

The reference's patch encoder is an un-vendored, frozen ConvMAE run through torch
(`save_latent.py:42-60`); ``BASELINE.json`` configs[1] names a ResNet-18 trained
end-to-end instead.  This module owns the parameters (torchvision names, conv
weights held as ``channels_last`` OIHW tensors == the kernels' [O][Kh][Kw][I]
layout) and a hand-scheduled forward/backward over ``libisic_hip.so``: implicit-GEMM
convolutions (forward + data gradient), transposing-LDS weight gradients,
BatchNorm statistics/apply/backward, pooling.  torch only provides device memory
and the autograd edge to the MIL head.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .lib import ERR_UNSUPPORTED, IsicHipError, call

LAYERS = ((64, 1), (128, 2), (256, 2), (512, 2))
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
# Partial rows of the conv-epilogue BatchNorm statistics.  The producing kernels are persistent (grid <= CU count = 256;
# the stem launches up to 1024 blocks) and add a block's sums into row `block % slots`: with at least as many rows as
# blocks every block owns its row, and `isic_bn_finalize` adds the rows in a fixed order -- the statistics, and with them
# the whole forward, are bit-reproducible from run to run (round 3; 32 shared rows made them depend on arrival order).
STAT_SLOTS = 256
STEM_STAT_SLOTS = 1024
_BF16 = torch.bfloat16


def _empty(shape, like, dtype=_BF16):
    return torch.empty(shape, device=like.device, dtype=dtype)


class _ConvSpec:
    __slots__ = ("name", "cin", "cout", "k", "stride", "pad")

    def __init__(self, name, cin, cout, k, stride, pad):
        self.name, self.cin, self.cout, self.k, self.stride, self.pad = name, cin, cout, k, stride, pad


class ResNet18Encoder(nn.Module):
    """``forward(images[N,3,H,W]) -> features[N,512]`` (fp32), train-mode BatchNorm
    statistics over the local batch."""

    def __init__(self, in_ch=3, layers=LAYERS):
        super().__init__()
        self.layers_cfg = tuple(layers)
        self.out_dim = self.layers_cfg[-1][0]
        if in_ch > 4:
            raise ValueError("stem kernel supports at most 4 input channels")
        self.in_ch = in_ch
        self.specs = {}

        def conv(name, cin, cout, k, stride, pad):
            w = torch.empty(cout, cin, k, k).contiguous(memory_format=torch.channels_last)
            nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
            self._register(name + ".weight", nn.Parameter(w))
            self.specs[name] = _ConvSpec(name, cin, cout, k, stride, pad)

        def bn(name, c):
            self._register(name + ".weight", nn.Parameter(torch.ones(c)))
            self._register(name + ".bias", nn.Parameter(torch.zeros(c)))
            self._register(name + ".running_mean", torch.zeros(c), buffer=True)
            self._register(name + ".running_var", torch.ones(c), buffer=True)
            self._register(name + ".num_batches_tracked", torch.zeros((), dtype=torch.long), buffer=True)

        conv("conv1", in_ch, 64, 7, 2, 3)
        bn("bn1", 64)
        inp = 64
        self.blocks = []
        for li, (planes, stride) in enumerate(self.layers_cfg, start=1):
            for b in range(2):
                pre = f"layer{li}.{b}"
                st = stride if b == 0 else 1
                conv(f"{pre}.conv1", inp, planes, 3, st, 1)
                bn(f"{pre}.bn1", planes)
                conv(f"{pre}.conv2", planes, planes, 3, 1, 1)
                bn(f"{pre}.bn2", planes)
                ds = st != 1 or inp != planes
                if ds:
                    conv(f"{pre}.downsample.0", inp, planes, 1, st, 0)
                    bn(f"{pre}.downsample.1", planes)
                self.blocks.append((pre, ds))
                inp = planes
        self._wcache = {}      # conv name -> (w_fwd bf16, w_dgrad bf16)
        self._wgrad_ws = None
        self._zeros = None            # fp64 arena for the per-layer statistics accumulators: ONE memset per pass
        self._zeros_used = 0
        self._side = None             # second HIP stream: weight gradients run beside the data-gradient chain
        self.wgrad_stream = False     # opt-in (attribute, not an environment switch): see _side_stream
        # BatchNorm-backward sums inside the producing data gradient (stages 2-4; conv_halo.hip STATS 2).  Same-box A/B at
        # 2048 images: BatchNorm passes -1.0 ms, data gradients +0.75 ms per step -> +0.6 % bags/s, inside the run-to-run
        # noise, while the convolution entry's roofline fraction drops 0.369 -> 0.352: off by default, kept as an option.
        self.fuse_bn_backward = False
        # the gradient through an identity block's skip is joined from (d out, ReLU mask) inside conv1's data gradient instead
        # of being written by bn2's backward and read back as an addend (bit-identical; A/B attribute)
        self.mask_identity_gradient = True
        # downsample blocks: both stride-2 data gradients in one launch (A/B attribute; one bf16 rounding less than two launches)
        self.pair_downsample_gradient = True
        self.fold_shortcut_norm = True      # downsample blocks: the shortcut's BatchNorm output is not materialised (training)
        self._tape_fused = False      # fusion mode of the tape being replayed (recorded at forward time)
        self.grad_ready_hook = None   # callable(list_of_param_names) fired as gradients complete (DDP overlap)

    # parameters are registered under dotted torchvision names via nested holder modules
    def _register(self, dotted, tensor, buffer=False):
        parts = dotted.split(".")
        mod = self
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        if buffer:
            mod.register_buffer(parts[-1], tensor)
        else:
            mod.register_parameter(parts[-1], tensor)

    def _get(self, dotted):
        mod = self
        for p in dotted.split("."):
            mod = getattr(mod, p)
        return mod

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        # .to()/.cuda() must keep conv weights in channels_last memory (= [O][Kh][Kw][I])
        for name in self.specs:
            p = self._get(name + ".weight")
            if not p.data.is_contiguous(memory_format=torch.channels_last):
                p.data = p.data.contiguous(memory_format=torch.channels_last)
        self._wcache = {}
        return out

    # ------------------------------------------------------------------ weights in kernel layouts
    def prepare_weights(self):
        """fp32 master weights -> bf16 kernel layouts (forward [O][Kh][Kw][I], data-gradient
        [I][Kh][Kw][O] flipped, packed stem).  Run once per forward: the optimizer updates the
        master copy through raw pointers, so no version counter can be trusted."""
        for name, sp in self.specs.items():
            p = self._get(name + ".weight")
            if not p.is_cuda:
                raise IsicHipError("ResNet18Encoder runs on the MI355X only (no CPU fallback)")
            if not p.data.is_contiguous(memory_format=torch.channels_last):
                p.data = p.data.contiguous(memory_format=torch.channels_last)
            ent = self._wcache.get(name)
            if name == "conv1":
                ws = ent[0] if ent else torch.empty(64 * 7 * 8 * 4, device=p.device, dtype=_BF16)
                call("isic_conv_stem_pack_bf16", p.data, ws)
                self._wcache[name] = (ws, None)
            else:
                n = sp.cout * sp.cin * sp.k * sp.k
                wf = ent[0] if ent else torch.empty(n, device=p.device, dtype=_BF16)
                wd = ent[1] if ent else torch.empty(n, device=p.device, dtype=_BF16)
                call("isic_conv_weight_prep_bf16", p.data, wf, wd, sp.cout, sp.cin, sp.k, sp.k)
                self._wcache[name] = (wf, wd)

    def _weights(self, name, need_dgrad):
        return self._wcache[name]

    # ------------------------------------------------------------------ primitive launches
    def _conv_fwd(self, x, name, with_stats=True):
        """conv forward; in training also returns the fused per-channel (sum, sumsq) partials of the
        rounded output (BatchNorm statistics without a second pass over it)."""
        sp = self.specs[name]
        N, H, W, C = x.shape
        Ho = (H + 2 * sp.pad - sp.k) // sp.stride + 1
        Wo = (W + 2 * sp.pad - sp.k) // sp.stride + 1
        wf, _ = self._weights(name, False)
        out = _empty((N, Ho, Wo, sp.cout), x)
        acc = None
        if with_stats and self.training:
            acc = self._zeros64((2, STAT_SLOTS, sp.cout), x.device)
        call("isic_conv2d_igemm_bf16", x, wf, out, N, H, W, C, Ho, Wo, sp.cout, sp.k, sp.k, sp.stride, 1, sp.pad, None,
             acc[0] if acc is not None else None, acc[1] if acc is not None else None, STAT_SLOTS)
        return out, acc

    def _conv_dgrad(self, dy, name, in_shape, addend=None, addend_mask=None):
        """d_x = dgrad(dy) (+ addend, joined in fp32 inside the kernel's epilogue; with ``addend_mask`` -- a 1-bit ReLU mask --
        the addend counts only where its bit is set: the gradient through a block's identity, never materialised)."""
        sp = self.specs[name]
        N, H, W, C = in_shape
        _, Ho, Wo, Co = dy.shape
        _, wd = self._weights(name, True)
        dx = _empty(in_shape, dy)
        if addend_mask is not None:
            call("isic_conv2d_igemm_maskadd_bf16", dy, wd, dx, N, Ho, Wo, Co, H, W, C, sp.k, sp.k, 1, sp.stride,
                 sp.k - 1 - sp.pad, addend, addend_mask)
            return dx
        call("isic_conv2d_igemm_bf16", dy, wd, dx, N, Ho, Wo, Co, H, W, C, sp.k, sp.k, 1, sp.stride, sp.k - 1 - sp.pad,
             addend, None, None, 0)
        return dx

    def _conv_dgrad_pair(self, dy, name, dy2, name2, in_shape):
        """d_x of a downsample block: dgrad(conv1 3x3 / 2)(dy) + dgrad(downsample 1x1 / 2)(dy2) in ONE launch
        (isic_conv2d_dgrad_pair_bf16: the 1x1 term accumulates in the even-pixel class of the 3x3 gradient; the two-launch
        form wrote an input-sized tensor that is three quarters zeros and read it back as an addend)."""
        sp, sp2 = self.specs[name], self.specs[name2]
        N, H, W, C = in_shape
        _, Ho, Wo, Co = dy.shape
        if self.pair_downsample_gradient and sp.k == 3 and sp.stride == 2 and sp.pad == 1 and sp2.k == 1 and sp2.stride == 2 \
                and sp2.pad == 0 and tuple(dy2.shape) == tuple(dy.shape):
            _, wd = self._weights(name, True)
            _, wd2 = self._weights(name2, True)
            dx = _empty(in_shape, dy)
            try:
                call("isic_conv2d_dgrad_pair_bf16", dy, wd, dy2, wd2, dx, N, Ho, Wo, Co, H, W, C)
                return dx
            except IsicHipError as e:
                if e.code != ERR_UNSUPPORTED:
                    raise
        dx2 = self._conv_dgrad(dy2, name2, in_shape)
        return self._conv_dgrad(dy, name, in_shape, addend=dx2)

    def _maskadd_ok(self, dy_shape, name, in_shape):
        """Can the data gradient of this layer take its addend masked on the fly (3x3 stride-1 layers on the
        pixels-staged-once kernels)?"""
        sp = self.specs[name]
        N, H, W, C = in_shape
        _, Ho, Wo, Co = dy_shape
        return bool(call("isic_conv2d_maskadd_supported", N, Ho, Wo, Co, H, W, C, sp.k, sp.k, 1, sp.stride, sp.k - 1 - sp.pad))

    def _dgrad_bnbwd_ok(self, dy_shape, name, in_shape):
        """Is there a fused kernel for `dgrad(dy) -> ReLU mask -> BatchNorm-backward sums` of this layer?  Decided by
        the mode the TAPE was recorded in (``_tape_fused``, set by ``run_backward`` from the tape), not by the live
        attribute: toggling ``fuse_bn_backward`` between forward and backward must not change the path half-way."""
        if not self._tape_fused:
            return False
        sp = self.specs[name]
        N, H, W, C = in_shape
        _, Ho, Wo, Co = dy_shape
        return bool(call("isic_conv2d_dgrad_bnbwd_supported", N, Ho, Wo, Co, H, W, C, sp.k, sp.k, 1, sp.stride,
                         sp.k - 1 - sp.pad))

    def _conv_dgrad_bnbwd(self, dy, name, in_shape, mask, yraw, addend=None):
        """dz = relu_mask ? dgrad(dy) (+ addend) : 0 and the per-slot sums (sum dz, sum dz * yraw) of the BatchNorm
        whose output the convolution consumed -- its backward then needs no reduction pass (conv_halo.hip, STATS 2)."""
        sp = self.specs[name]
        N, H, W, C = in_shape
        _, Ho, Wo, Co = dy.shape
        _, wd = self._weights(name, True)
        dz = _empty(in_shape, dy)
        sums = self._zeros64((2, STAT_SLOTS, C), dy.device)
        call("isic_conv2d_dgrad_bnbwd_bf16", dy, wd, dz, N, Ho, Wo, Co, H, W, C, sp.k, sp.k, 1, sp.stride, sp.k - 1 - sp.pad,
             addend, mask, yraw, sums[0], sums[1], STAT_SLOTS)
        return dz, sums

    def _bn_bwd_from_sums(self, dz, c, st, name, sums):
        """BatchNorm backward when dz (ReLU mask already applied) and its sums came out of the producing data gradient."""
        mean, rstd = st[0], st[1]
        N, H, W, C = c.shape
        gamma, beta = self._get(name + ".weight"), self._get(name + ".bias")
        acc = self._zeros64((2, C), c.device)
        call("isic_bn_bwd_finalize", sums[0], sums[1], STAT_SLOTS, C, mean, rstd, acc[0], acc[1])
        dx = _empty(c.shape, c)
        call("isic_bn_bwd_apply_bf16", dz, c, None, mean, rstd, gamma.data, acc[0], acc[1], N * H * W, C, 0, None, None, dx,
             None, self._grad_buffer(gamma), self._grad_buffer(beta))
        return dx

    def _grad_buffer(self, p):
        if p.grad is None:
            p.grad = torch.zeros_like(p.data, memory_format=torch.preserve_format)
        return p.grad

    _ARENA_DOUBLES = 2 * STAT_SLOTS * 4800 + 2 * 64 * STEM_STAT_SLOTS + 8192     # all conv / BN layers of ResNet-18, one pass

    def _arena_reset(self, device):
        """Zero the statistics arena (one fill kernel instead of one torch.zeros per layer)."""
        if self._zeros is None or self._zeros.device != device:
            self._zeros = torch.empty(self._ARENA_DOUBLES, device=device, dtype=torch.float64)
        self._zeros.zero_()
        self._zeros_used = 0

    def _zeros64(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        if self._zeros is None or self._zeros.device != device or self._zeros_used + n > self._zeros.numel():
            return torch.zeros(shape, device=device, dtype=torch.float64)   # outside a pass / arena exhausted
        v = self._zeros[self._zeros_used:self._zeros_used + n].view(shape)
        self._zeros_used += n
        return v

    def _side_stream(self, device):
        """Weight gradients are leaves of the backward graph: with ``self.wgrad_stream = True`` they run on a second
        stream, so that the MFMA-bound wgrad kernels overlap the HBM-bound BatchNorm passes of the
        data-gradient chain and fill the partial last round of its convolution grids (+2 % bags/s on one
        MI355X).  Off by default: concurrent kernels stretch each other, which blurs per-kernel timings
        (bench.py's roofline, rocprofv3 summaries)."""
        if not self.wgrad_stream:
            return None
        if self._side is None or self._side.device != device:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def _on_side(self, device, tensors, fn):
        """Run ``fn`` (kernel launches reading ``tensors``, produced on the current stream) on the side stream."""
        side = self._side_stream(device)
        if side is None:
            fn()
            return
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            fn()
        for t in tensors:
            t.record_stream(side)     # the caching allocator must not hand the block out before the side work is done

    def _workspace(self, nbytes, device):
        """The weight-gradient workspace (source-offset table + split-K / per-block partials): one buffer, grown to the
        largest request; the launches that use it are ordered on one stream (the side stream when it is on)."""
        ws = self._wgrad_ws
        if ws is None or ws.numel() < nbytes or ws.device != device:
            ws = self._wgrad_ws = torch.empty(int(nbytes), device=device, dtype=torch.uint8)   # caching allocator: 512-B aligned
        return ws

    def _conv_wgrad(self, x, dy, name):
        self._on_side(x.device, (x, dy), lambda: self._conv_wgrad_now(x, dy, name))

    def _conv_wgrad_now(self, x, dy, name):
        sp = self.specs[name]
        p = self._get(name + ".weight")
        g = self._grad_buffer(p)
        if not g.is_contiguous(memory_format=torch.channels_last):
            raise IsicHipError(f"{name}.weight.grad must be channels_last ([O][Kh][Kw][I] memory)")
        N, H, W, C = x.shape
        _, Ho, Wo, Co = dy.shape
        ws = self._workspace(call("isic_conv2d_wgrad_workspace_bytes", N, C, Ho, Wo, Co, sp.k, sp.k), x.device)
        call("isic_conv2d_wgrad_bf16", x, dy, g, N, H, W, C, Ho, Wo, Co, sp.k, sp.k, sp.stride, sp.pad, ws, ws.numel())

    def _bn_fwd(self, c, name, relu, residual=None, acc=None, residual_affine=None):
        """Returns (y, (mean, rstd, scale, shift)).  ``acc`` = fused statistics from the producing conv.
        ``residual_affine`` = (scale, shift): ``residual`` is a RAW convolution output normalised on the fly (a downsample
        block's shortcut: its BatchNorm's output is never written)."""
        st = self._bn_affine(c, name, acc)
        N, H, W, C = c.shape
        y = _empty(c.shape, c)
        if residual_affine is not None:
            mask = torch.empty((N * H * W * C) // 8, device=c.device, dtype=torch.uint8)
            call("isic_bn_apply_mask_res_affine_bf16", c, st[2], st[3], residual, residual_affine[0], residual_affine[1], y, mask,
                 N * H * W, C)
            return y, st + (mask,)
        if relu and self.training and (residual is not None or self.fuse_bn_backward):
            # the backward passes need the ReLU mask of relu(bn(c) [+ residual]): kept as 1 bit per element (with a
            # residual it cannot be recomputed from c; without one it is what the fused data-gradient epilogue reads)
            mask = torch.empty((N * H * W * C) // 8, device=c.device, dtype=torch.uint8)
            call("isic_bn_apply_mask_bf16", c, st[2], st[3], residual, y, mask, N * H * W, C)
            return y, st + (mask,)
        call("isic_bn_apply_bf16", c, st[2], st[3], residual, y, N * H * W, C, int(relu))
        return y, st

    def _bn_affine(self, c, name, acc=None):
        """Batch statistics (train) or running statistics (eval) -> (mean, rstd, scale, shift) of y = c*scale + shift."""
        N, H, W, C = c.shape
        rows = N * H * W
        gamma, beta = self._get(name + ".weight"), self._get(name + ".bias")
        dev = c.device
        scale = torch.empty(C, device=dev, dtype=torch.float32)
        shift = torch.empty(C, device=dev, dtype=torch.float32)
        mean = rstd = None
        if self.training:
            slots = int(acc.shape[1]) if acc is not None else 1
            if acc is None:
                slots = 1
                acc = self._zeros64((2, 1, C), dev)
                call("isic_bn_stats_bf16", c, rows, C, acc[0], acc[1])
            mean = torch.empty(C, device=dev, dtype=torch.float32)
            rstd = torch.empty(C, device=dev, dtype=torch.float32)
            call("isic_bn_finalize", acc[0], acc[1], slots, rows, C, gamma.data, beta.data, BN_EPS, BN_MOMENTUM, scale,
                 shift, mean, rstd, self._get(name + ".running_mean"), self._get(name + ".running_var"))
        else:
            call("isic_bn_eval_affine", gamma.data, beta.data, self._get(name + ".running_mean"),
                 self._get(name + ".running_var"), BN_EPS, C, scale, shift)
        return mean, rstd, scale, shift

    def _bn_bwd(self, dy, c, y, st, name, relu, want_residual, mask_from_x=False):
        """BatchNorm(+ReLU) backward.  ``mask_from_x``: no residual was added, so the ReLU mask is
        recomputed from c*scale+shift and y is not read."""
        mean, rstd, scale, shift = st[:4]
        mask = st[4] if len(st) > 4 else None
        N, H, W, C = c.shape
        rows = N * H * W
        gamma, beta = self._get(name + ".weight"), self._get(name + ".bias")
        acc = self._zeros64((2, C), c.device)
        if mask is not None and relu and not mask_from_x:
            call("isic_bn_bwd_reduce_mask_bf16", dy, c, mask, mean, rstd, rows, C, acc[0], acc[1])
            dx = _empty(c.shape, c)
            dres = _empty(c.shape, c) if want_residual else None
            call("isic_bn_bwd_apply_mask_bf16", dy, c, mask, mean, rstd, gamma.data, acc[0], acc[1], rows, C, dx, dres,
                 self._grad_buffer(gamma), self._grad_buffer(beta))
            return dx, dres
        sc, sh = (scale, shift) if (mask_from_x and relu) else (None, None)
        yy = None if (mask_from_x or not relu) else y
        call("isic_bn_bwd_reduce_bf16", dy, c, yy, mean, rstd, rows, C, int(relu), sc, sh, acc[0], acc[1])
        dx = _empty(c.shape, c)
        dres = _empty(c.shape, c) if want_residual else None
        call("isic_bn_bwd_apply_bf16", dy, c, yy, mean, rstd, gamma.data, acc[0], acc[1], rows, C, int(relu), sc, sh, dx,
             dres, self._grad_buffer(gamma), self._grad_buffer(beta))
        return dx, dres

    def _fire(self, names):
        if self.grad_ready_hook is not None:
            # the hook hands gradients to the collective stream, which waits for the stream it is called on:
            # the side stream, once it has caught up with everything the main stream produced so far
            dev = self._get("conv1.weight").device
            self._on_side(dev, (), lambda: self.grad_ready_hook(names))

    # ------------------------------------------------------------------ forward / backward
    def pack_input(self, images):
        """NCHW fp32/bf16 images -> NHWC bf16 with C padded to 4."""
        if images.dim() != 4 or images.shape[1] != self.in_ch:
            raise ValueError(f"expected images[N,{self.in_ch},H,W], got {tuple(images.shape)}")
        if not images.is_cuda:
            raise IsicHipError("ResNet18Encoder runs on the MI355X only (no CPU fallback)")
        if images.dtype not in (torch.float32, _BF16):
            images = images.float()
        images = images.contiguous()
        N, C, H, W = images.shape
        out = torch.empty((N, H, W, 4), device=images.device, dtype=_BF16)
        call("isic_nchw_to_nhwc4_bf16", images, int(images.dtype == _BF16), out, N, C, H, W)
        return out

    def block_forward(self, x, pre, ds):
        """One BasicBlock: relu(bn2(conv2(relu(bn1(conv1(x))))) + identity), identity = x or
        bn(conv1x1/2(x)).  ``x`` NHWC bf16.  Returns (out, saved-for-backward)."""
        idn, cd, std, idn_affine = x, None, None, None
        if ds:
            cd, accd = self._conv_fwd(x, f"{pre}.downsample.0")
            if self.training and self.fold_shortcut_norm:
                # the shortcut's BatchNorm (no ReLU) is applied where it is consumed -- inside bn2's apply pass: one write
                # and one read of a block-output-sized tensor less, bit-identical (the inner value is rounded as it was)
                std = self._bn_affine(cd, f"{pre}.downsample.1", accd)
                idn, idn_affine = cd, (std[2], std[3])
            else:
                idn, std = self._bn_fwd(cd, f"{pre}.downsample.1", False, acc=accd)
        c1, acc1 = self._conv_fwd(x, f"{pre}.conv1")
        a1, st1 = self._bn_fwd(c1, f"{pre}.bn1", True, acc=acc1)
        c2, acc2 = self._conv_fwd(a1, f"{pre}.conv2")
        out, st2 = self._bn_fwd(c2, f"{pre}.bn2", True, residual=idn, acc=acc2, residual_affine=idn_affine)
        return out, (x, c1, a1, st1, c2, out, st2, cd, std)

    def block_backward(self, g, pre, ds, saved):
        """Backward of ``block_forward``: ``g`` = d loss / d out (NHWC bf16).  Accumulates the block's parameter
        gradients into ``param.grad`` and returns (d loss / d x, names of the parameters whose gradients are final)."""
        dx, names, _ = self.block_backward_fused(g, pre, ds, saved)
        return dx, names

    def block_backward_fused(self, g, pre, ds, saved, g_sums=None, prev=None):
        """``block_backward`` with the BatchNorm-backward reductions folded into the data gradients that produce their
        inputs, where a fused kernel exists (3x3 stride-1 layers with >= 128 channels):
          * ``g_sums`` given: ``g`` is ALREADY dz of bn2 (ReLU mask applied by the producer) with its (sum dz, sum dz*c2);
          * ``prev`` = ``saved`` of the preceding block: the data gradient of conv1 then applies THAT block's output
            mask and returns dz of its bn2 together with the sums (third return value, else None)."""
        x, c1, a1, st1, c2, out, st2, cd, std = saved
        # identity block: the gradient through the identity is g where relu(bn2 + x) was active -- conv1's data gradient joins
        # it straight from (g, mask) (isic_conv2d_igemm_maskadd_bf16) and bn2's backward does not write it (round 4)
        masked_join = (self.mask_identity_gradient and not ds and g_sums is None and len(st2) > 4 and
                       not (prev is not None and len(prev[6]) > 4 and self._dgrad_bnbwd_ok(c1.shape, f"{pre}.conv1", tuple(x.shape)))
                       and self._maskadd_ok(c1.shape, f"{pre}.conv1", tuple(x.shape)))
        if g_sums is not None:
            dc2, dres = self._bn_bwd_from_sums(g, c2, st2, f"{pre}.bn2", g_sums), g      # the residual gradient IS dz
        else:
            dc2, dres = self._bn_bwd(g, c2, out, st2, f"{pre}.bn2", True, not masked_join)
        self._conv_wgrad(a1, dc2, f"{pre}.conv2")
        if len(st1) > 4 and self._dgrad_bnbwd_ok(dc2.shape, f"{pre}.conv2", tuple(a1.shape)):
            dz1, sums1 = self._conv_dgrad_bnbwd(dc2, f"{pre}.conv2", tuple(a1.shape), st1[4], c1)
            del dc2
            dc1 = self._bn_bwd_from_sums(dz1, c1, st1, f"{pre}.bn1", sums1)
            del dz1
        else:
            da1 = self._conv_dgrad(dc2, f"{pre}.conv2", tuple(a1.shape))
            del dc2
            dc1, _ = self._bn_bwd(da1, c1, a1, st1, f"{pre}.bn1", True, False, mask_from_x=True)
            del da1
        self._conv_wgrad(x, dc1, f"{pre}.conv1")
        names = [f"{pre}.conv2.weight", f"{pre}.bn2.weight", f"{pre}.bn2.bias", f"{pre}.conv1.weight",
                 f"{pre}.bn1.weight", f"{pre}.bn1.bias"]
        dx_sums = None
        if ds:
            dcd, _ = self._bn_bwd(dres, cd, None, std, f"{pre}.downsample.1", False, False)
            self._conv_wgrad(x, dcd, f"{pre}.downsample.0")
            dx = self._conv_dgrad_pair(dc1, f"{pre}.conv1", dcd, f"{pre}.downsample.0", tuple(x.shape))
            names += [f"{pre}.downsample.0.weight", f"{pre}.downsample.1.weight", f"{pre}.downsample.1.bias"]
        elif prev is not None and len(prev[6]) > 4 and self._dgrad_bnbwd_ok(dc1.shape, f"{pre}.conv1", tuple(x.shape)):
            dx, dx_sums = self._conv_dgrad_bnbwd(dc1, f"{pre}.conv1", tuple(x.shape), prev[6][4], prev[4], addend=dres)
        elif masked_join:
            dx = self._conv_dgrad(dc1, f"{pre}.conv1", tuple(x.shape), addend=g, addend_mask=st2[4])
        else:
            dx = self._conv_dgrad(dc1, f"{pre}.conv1", tuple(x.shape), addend=dres)   # + identity gradient
        return dx, names, dx_sums

    def run_forward(self, images, save):
        """Returns (features[N,512] fp32, tape).  ``save=False`` drops everything not
        needed (inference)."""
        self.prepare_weights()
        x0 = self.pack_input(images)
        self._arena_reset(x0.device)
        N, H, W, _ = x0.shape
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        ws, _ = self._weights("conv1", False)
        c = _empty((N, Ho, Wo, 64), x0)
        acc0 = None
        if self.training:      # batch statistics of bn1 fused into the stem convolution's epilogue
            acc0 = self._zeros64((2, STEM_STAT_SLOTS, 64), x0.device)
            call("isic_conv_stem_fwd_stats_bf16", x0, ws, c, N, H, W, Ho, Wo, acc0[0], acc0[1], STEM_STAT_SLOTS)
        else:
            call("isic_conv_stem_fwd_bf16", x0, ws, c, N, H, W, Ho, Wo)
        # bn1 + relu + maxpool in one pass over the stem activation (the largest tensor of the network): the
        # normalised tensor is never written; backward recomputes the ReLU mask from c
        st0 = self._bn_affine(c, "bn1", acc0)
        Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
        p = _empty((N, Hp, Wp, 64), c)
        am = torch.empty((N, Hp, Wp, 64), device=c.device, dtype=torch.uint8) if save else None
        csel = _empty((N, Hp, Wp, 64), c) if save else None    # raw stem output at each window's argmax (bn1 backward sums)
        call("isic_bn_relu_maxpool3x3s2_fwd_sel_bf16", c, st0[2], st0[3], p, am, csel, N, Ho, Wo, 64, Hp, Wp)
        tape = {"x0": x0, "stem": (c, st0, am, csel, (N, Ho, Wo, 64)), "blocks": [],
                "fused_bn_backward": bool(self.fuse_bn_backward)} if save else None
        x = p
        for pre, ds in self.blocks:
            x, saved = self.block_forward(x, pre, ds)
            if save:
                tape["blocks"].append(saved)
        N, Hf, Wf, Cf = x.shape
        feat = torch.empty((N, Cf), device=x.device, dtype=torch.float32)
        call("isic_avgpool_fwd_bf16", x, feat, N, Hf * Wf, Cf)
        if save:
            tape["final_shape"] = (N, Hf, Wf, Cf)
            if self.training:
                for name in self._bn_names():
                    self._get(name + ".num_batches_tracked").add_(1)
        return feat, tape

    @torch.no_grad()
    def run_tokens(self, images, stage=3):
        """Frozen-encoder token grid of the latent extraction path (`save_latent.py:53-60`): the activation after
        residual stage ``stage`` as ``[N, h*w, C]`` fp32 tokens.  At stage 3 a 224x224 image gives 14x14 = 196 tokens of
        256 channels, one per 16x16-pixel patch -- the geometry of the reference's ViT latents (`save_latent.py:77`)."""
        if self.training:
            raise IsicHipError("run_tokens is an inference path: call eval() first (running BatchNorm statistics)")
        self.prepare_weights()
        x0 = self.pack_input(images)
        self._arena_reset(x0.device)
        N, H, W, _ = x0.shape
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        ws, _ = self._weights("conv1", False)
        c = _empty((N, Ho, Wo, 64), x0)
        call("isic_conv_stem_fwd_bf16", x0, ws, c, N, H, W, Ho, Wo)
        st0 = self._bn_affine(c, "bn1", None)
        Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
        x = _empty((N, Hp, Wp, 64), c)
        call("isic_bn_relu_maxpool3x3s2_fwd_bf16", c, st0[2], st0[3], x, None, N, Ho, Wo, 64, Hp, Wp)
        for pre, ds in self.blocks[:2 * int(stage)]:
            x, _ = self.block_forward(x, pre, ds)
        N, Hf, Wf, Cf = x.shape
        return x.float().view(N, Hf * Wf, Cf)

    def _bn_names(self):
        names = ["bn1"]
        for pre, ds in self.blocks:
            names += [f"{pre}.bn1", f"{pre}.bn2"] + ([f"{pre}.downsample.1"] if ds else [])
        return names

    def run_backward(self, tape, dfeat):
        """Accumulates every parameter gradient (into ``param.grad``) from d loss / d features."""
        if not self.training:
            raise IsicHipError("encoder backward needs train() mode (batch-statistics BatchNorm)")
        N, Hf, Wf, Cf = tape["final_shape"]
        self._tape_fused = bool(tape.get("fused_bn_backward", False))
        self._arena_reset(dfeat.device)
        g = torch.empty((N, Hf, Wf, Cf), device=dfeat.device, dtype=_BF16)
        call("isic_avgpool_bwd_bf16", dfeat.float().contiguous(), g, N, Hf * Wf, Cf)
        g_sums = None
        for bi in range(len(self.blocks) - 1, -1, -1):
            pre, ds = self.blocks[bi]
            prev = tape["blocks"][bi - 1] if bi > 0 else None
            g, names, g_sums = self.block_backward_fused(g, pre, ds, tape["blocks"][bi], g_sums=g_sums, prev=prev)
            self._fire(names)
        c, st0, am, csel, yshape = tape["stem"]
        N, Ho, Wo, _ = yshape
        _, Hp, Wp, _ = g.shape
        # bn1 backward without a full-size gradient tensor.  Sums: a pooling window's gradient reaches exactly its
        # argmax pixel, so they are a pass over the POOLED tensors (g, the raw activation at the argmax).  Apply: fused
        # into the stem weight gradient, which forms dY = bn_bwd(maxpool_bwd(g)) in registers from c, g and argmax.
        mean, rstd, scale, shift = st0
        gamma, beta = self._get("bn1.weight"), self._get("bn1.bias")
        acc = self._zeros64((2, 64), c.device)
        call("isic_bn_bwd_reduce_bf16", g, csel, None, mean, rstd, N * Hp * Wp, 64, 1, scale, shift, acc[0], acc[1])
        x0 = tape["x0"]
        p = self._get("conv1.weight")
        gw, gg, gb = self._grad_buffer(p), self._grad_buffer(gamma), self._grad_buffer(beta)
        ws = self._workspace(call("isic_conv_stem_wgrad_workspace_bytes"), c.device)
        self._on_side(c.device, (x0, c, g, am, acc),
                      lambda: call("isic_conv_stem_wgrad_bn_pooled_bf16", x0, c, am, g, mean, rstd, gamma.data, scale, shift,
                                   acc[0], acc[1], gw, gg, gb, N, x0.shape[1], x0.shape[2], Ho, Wo, Hp, Wp, ws, ws.numel()))
        self._fire(["conv1.weight", "bn1.weight", "bn1.bias"])
        if self._side is not None:
            torch.cuda.current_stream(c.device).wait_stream(self._side)    # gradients complete for whoever comes next

    def forward(self, images):
        if torch.is_grad_enabled() and self.training:
            return _EncoderFn.apply(images, self, *[p for p in self.parameters()])
        feat, _ = self.run_forward(images, save=False)
        return feat


class _EncoderFn(torch.autograd.Function):
    """Autograd edge: features -> encoder parameter gradients.  The parameters are
    passed as inputs only so that autograd schedules this node; their gradients are
    accumulated in place by the kernels (``param.grad``), hence ``None`` is returned."""

    @staticmethod
    def forward(ctx, images, enc, *params):
        feat, tape = enc.run_forward(images, save=True)
        ctx.enc, ctx.tape = enc, tape
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        enc, tape = ctx.enc, ctx.tape
        ctx.tape = None
        enc.run_backward(tape, dfeat)
        return (None, None) + tuple(None for _ in enc.parameters())
