"""ViT-S/16 patch encoder, fp16 on MFMA -- the frozen encoder of BASELINE.json configs[4].

The reference's patch encoder is an un-vendored ConvMAE conv-ViT run frozen: ``model.eval()``, ``torch.no_grad()``,
``forward(images, mask_ratio=0) -> latent[B, 196, 768]`` (`save_latent.py:42-60`).  Its code and weights are not in the
tree, so this is the build's own ViT-S/16 (timm `vit_small_patch16_224` parameter names, no class token: the reference
consumes the 196 patch tokens only, `save_latent.py:56-60,77`).  Inference only, like the reference's use of it.

Every layer is one launch of the C ABI: `isic_vit_patchify_f16`, `isic_gemm_f16` (bias, GELU, residual / position
embedding fused into the epilogue), `isic_layernorm_f16`, `isic_attention_f16`.  The residual stream and all
activations are fp16 in HBM ([N*196, 384] rows), arithmetic is fp32 inside the kernels.  No CPU fallback.

``fold_layernorm`` (default True, round 3): the two pre-norm LayerNorms of a block have no pass of their own.  With
W' = W diag(gamma), c = W' 1, b' = b + W beta:   LN(x) W^T + b = rstd (x W'^T - mean c) + b'   -- the product reads the RAW
residual stream (`isic_gemm_f16_ln`) and applies the row's (mean, rstd) in its epilogue; the row sums those need come
out of the epilogue of the product that WROTE the stream (`isic_gemm_f16_stats`: patch projection, attn.proj, mlp.fc2).
``"stats"`` keeps a statistics-only pass (`isic_row_stats_f16`, LayerNorm's two-pass arithmetic) in place of the epilogue
sums; ``False`` is the layer-by-layer form above.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .lib import IsicHipError, call

_F16 = torch.float16


class ViTSmallEncoder(nn.Module):
    def __init__(self, img_size=224, patch=16, in_ch=3, dim=384, depth=12, heads=6, mlp_ratio=4, seed=0,
                 fold_layernorm=True):
        super().__init__()
        if fold_layernorm not in (True, False, "stats"):
            raise ValueError("fold_layernorm: True, False or 'stats'")
        self.fold_layernorm = fold_layernorm
        if dim % 128 != 0 or dim // heads != 64 or patch % 8 != 0 or img_size % patch != 0:
            raise ValueError("ViTSmallEncoder: dim % 128 == 0, head width 64, patch % 8 == 0, img_size % patch == 0")
        self.img_size, self.patch, self.in_ch, self.dim, self.depth, self.heads = img_size, patch, in_ch, dim, depth, heads
        self.mlp = dim * mlp_ratio
        self.tokens = (img_size // patch) ** 2
        if self.tokens > 208:
            raise ValueError("ViTSmallEncoder: at most 208 tokens per image (attention kernel)")
        self.feature_dim = dim
        g = torch.Generator().manual_seed(seed)

        def P(*shape, scale):
            return nn.Parameter(torch.randn(*shape, generator=g) * scale)
        self._names = []

        def add(name, param):
            self._names.append(name)
            self.register_parameter(name.replace(".", "__"), param)
        add("patch_embed.proj.weight", P(dim, in_ch, patch, patch, scale=1.0 / math.sqrt(in_ch * patch * patch)))
        add("patch_embed.proj.bias", nn.Parameter(torch.zeros(dim)))
        add("pos_embed", P(1, self.tokens, dim, scale=0.02))
        for i in range(depth):
            b = f"blocks.{i}"
            add(f"{b}.norm1.weight", nn.Parameter(torch.ones(dim))); add(f"{b}.norm1.bias", nn.Parameter(torch.zeros(dim)))
            add(f"{b}.attn.qkv.weight", P(3 * dim, dim, scale=0.02)); add(f"{b}.attn.qkv.bias", nn.Parameter(torch.zeros(3 * dim)))
            add(f"{b}.attn.proj.weight", P(dim, dim, scale=0.02)); add(f"{b}.attn.proj.bias", nn.Parameter(torch.zeros(dim)))
            add(f"{b}.norm2.weight", nn.Parameter(torch.ones(dim))); add(f"{b}.norm2.bias", nn.Parameter(torch.zeros(dim)))
            add(f"{b}.mlp.fc1.weight", P(self.mlp, dim, scale=0.02)); add(f"{b}.mlp.fc1.bias", nn.Parameter(torch.zeros(self.mlp)))
            add(f"{b}.mlp.fc2.weight", P(dim, self.mlp, scale=0.02)); add(f"{b}.mlp.fc2.bias", nn.Parameter(torch.zeros(dim)))
        add("norm.weight", nn.Parameter(torch.ones(dim))); add("norm.bias", nn.Parameter(torch.zeros(dim)))
        for p in self.parameters():
            p.requires_grad_(False)                    # frozen, as in save_latent.py:51-53
        self._w16 = None                               # fp16 copies of the matrices, made once per weight version
        self._w16_key = None
        self.eval()

    # ------------------------------------------------------------------ timm-named state_dict
    def _get(self, name):
        return self._parameters[name.replace(".", "__")]

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else {}
        for n in self._names:
            p = self._get(n)
            out[prefix + n] = p if keep_vars else p.detach()
        return out

    def load_state_dict(self, state_dict, strict=True, assign=False):
        # a timm vit_small_patch16_224 checkpoint carries a class token: ``cls_token`` [1,1,D] and a ``pos_embed`` of
        # tokens + 1 rows whose row 0 is the class token's position.  This encoder has no class token (the reference
        # keeps the 196 patch latents, `save_latent.py:60`): row 0 is dropped and ``cls_token`` ignored.
        state_dict = dict(state_dict)
        state_dict.pop("cls_token", None)
        pe = state_dict.get("pos_embed")
        if pe is not None and pe.dim() == 3 and pe.shape[1] == self.tokens + 1:
            state_dict["pos_embed"] = pe[:, 1:, :]
        missing = [n for n in self._names if n not in state_dict]
        unexpected = [k for k in state_dict if k not in self._names]
        if strict and (missing or unexpected):
            raise RuntimeError(f"ViTSmallEncoder.load_state_dict: missing {missing[:4]}, unexpected {unexpected[:4]}")
        with torch.no_grad():
            for n in self._names:
                if n in state_dict:
                    src = state_dict[n]
                    dst = self._get(n)
                    if tuple(src.shape) != tuple(dst.shape):
                        raise RuntimeError(f"size mismatch for {n}: {tuple(src.shape)} vs {tuple(dst.shape)}")
                    dst.copy_(src)
        self._w16_key = None
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def train(self, mode=True):
        if mode:
            raise IsicHipError("ViTSmallEncoder is a frozen inference encoder (save_latent.py:51-53): no train() mode")
        return super().train(False)

    # ------------------------------------------------------------------ forward
    def _prepare(self, device):
        key = tuple((self._get(n).data_ptr(), self._get(n)._version) for n in self._names)
        if self._w16 is not None and key == self._w16_key:
            return self._w16
        w = {}
        for n in self._names:
            p = self._get(n).detach()
            if p.device != device:
                raise IsicHipError("ViTSmallEncoder: move the module to the GPU first (.to('cuda'))")
            if n == "patch_embed.proj.weight":
                w[n] = p.reshape(self.dim, -1).to(_F16).contiguous()
            elif n == "pos_embed":
                w[n] = p.reshape(self.tokens, self.dim).to(_F16).contiguous()
            elif n.endswith(".weight") and p.dim() == 2:
                w[n] = p.to(_F16).contiguous()
            else:
                w[n] = p.float().contiguous()                                      # biases, LayerNorm affine: fp32
        # LayerNorm folded into the product that follows it (module docstring): W' fp16, c from the ROUNDED W' (it has to
        # cancel what the MFMAs sum), b' fp32
        for i in range(self.depth):
            for norm, lin in ((f"blocks.{i}.norm1", f"blocks.{i}.attn.qkv"), (f"blocks.{i}.norm2", f"blocks.{i}.mlp.fc1")):
                W = self._get(lin + ".weight").detach().float()
                gamma, beta = w[norm + ".weight"], w[norm + ".bias"]
                Wg = (W * gamma[None, :]).to(_F16).contiguous()
                w[lin + ".ln_weight"] = Wg
                w[lin + ".ln_c"] = Wg.float().sum(dim=1).contiguous()
                w[lin + ".ln_bias"] = (w[lin + ".bias"] + W @ beta).contiguous()
        self._w16, self._w16_key = w, key
        return w

    @torch.no_grad()
    def run_tokens(self, images, depth=None):
        """images[N,3,H,W] (fp32; other float types are converted) on the GPU -> tokens[N, 196, 384] fp32."""
        if images.dim() != 4 or images.shape[1] != self.in_ch or images.shape[2] != self.img_size or images.shape[3] != self.img_size:
            raise ValueError(f"expected images[N,{self.in_ch},{self.img_size},{self.img_size}], got {tuple(images.shape)}")
        if not images.is_cuda:
            raise IsicHipError("ViTSmallEncoder runs on the MI355X only (no CPU fallback)")
        dev = images.device
        w = self._prepare(dev)
        x_in = images.float().contiguous()
        N, T, D, H = x_in.shape[0], self.tokens, self.dim, self.heads
        M = N * T
        K0 = self.in_ch * self.patch * self.patch
        rows = torch.empty((M, K0), device=dev, dtype=_F16)
        call("isic_vit_patchify_f16", x_in, rows, N, self.in_ch, self.img_size, self.img_size, self.patch)
        x = torch.empty((M, D), device=dev, dtype=_F16)
        fold = self.fold_layernorm
        eps = 1e-6
        h = torch.empty((M, D), device=dev, dtype=_F16) if fold is False else None
        qkv = torch.empty((M, 3 * D), device=dev, dtype=_F16)
        att = torch.empty((M, D), device=dev, dtype=_F16)
        hid = torch.empty((M, self.mlp), device=dev, dtype=_F16)
        x2 = torch.empty_like(x)
        # row statistics of x / x2: partial sums per 64-column group out of the producing epilogue (fold True), or
        # (mean, rstd) from a statistics-only pass ("stats")
        parts = 2 * D // 128 if fold is True else 0
        st = torch.empty((M, max(parts, 1), 2), device=dev, dtype=torch.float32) if fold is not False else None
        st2 = torch.empty_like(st) if fold is not False else None

        def linear_res(a, name, res, out, stats, K, res_rows=0):
            """out = a . W^T + b + res; the LayerNorm statistics of out on the way when the next product wants them"""
            if fold is True:
                call("isic_gemm_f16_stats", a, w[name + ".weight"], w[name + ".bias"], res, out, stats, M, D, K, 0, res_rows)
            else:
                call("isic_gemm_f16", a, w[name + ".weight"], w[name + ".bias"], res, out, M, D, K, 0, res_rows)
                if fold == "stats":
                    call("isic_row_stats_f16", out, stats, M, D, eps)

        def linear_ln(xin, stats, norm, name, out, Nout, act):
            """out = act(LayerNorm(xin) . W^T + b)"""
            if fold is False:
                call("isic_layernorm_f16", xin, w[norm + ".weight"], w[norm + ".bias"], h, None, M, D, eps)
                call("isic_gemm_f16", h, w[name + ".weight"], w[name + ".bias"], None, out, M, Nout, D, act, 0)
            else:
                call("isic_gemm_f16_ln", xin, w[name + ".ln_weight"], w[name + ".ln_bias"], w[name + ".ln_c"], stats, parts, out,
                     M, Nout, D, act, eps)

        linear_res(rows, "patch_embed.proj", w["pos_embed"], x, st, K0, res_rows=T)
        del rows
        for i in range(self.depth if depth is None else depth):
            b = f"blocks.{i}"
            linear_ln(x, st, f"{b}.norm1", f"{b}.attn.qkv", qkv, 3 * D, 0)
            call("isic_attention_f16", qkv, att, N, T, H, D // H)
            linear_res(att, f"{b}.attn.proj", x, x2, st2, D)
            linear_ln(x2, st2, f"{b}.norm2", f"{b}.mlp.fc1", hid, self.mlp, 1)
            linear_res(hid, f"{b}.mlp.fc2", x2, x, st, self.mlp)
        out = torch.empty((M, D), device=dev, dtype=torch.float32)
        call("isic_layernorm_f16", x, w["norm.weight"], w["norm.bias"], None, out, M, D, 1e-6)
        return out.view(N, T, D)

    def forward(self, images):
        """Mean-pooled 384-d feature per image (what a MIL bag of patches consumes)."""
        return self.run_tokens(images).mean(dim=1)

    def flops_per_image(self):
        T, D, Hm = self.tokens, self.dim, self.mlp
        per_block = 2 * T * (D * 3 * D + D * D + 2 * D * Hm) + 4 * T * T * D
        return 2 * T * D * self.in_ch * self.patch ** 2 + self.depth * per_block
