"""DINOv2 ViT backbone (ViT-L/14 with 4 registers) on HIP kernels.

Mirrors reference iggt/layers/vision_transformer.py:42-340,379-390 (parameter names, ctor
signature, returned dict).  The forward is re-designed for the MI355X:

  images fp32 [S,3,H,W] --(iggt_im2row_patch14: ImageNet-normalise + unfold -> bf16 [S*g2, 640])-->
  --(bf16 MFMA GEMM, epilogue: + conv bias + pos_embed[patch], rows scattered behind the 5 special
     tokens of each view)--> x fp32 [S, 5+g2, 1024]   (cls + pos_embed[0], 4 registers written by
     iggt_write_special_tokens) --> 24 blocks in place --> final LayerNorm of the patch rows.

NOTE the normalisation: the reference normalises in Aggregator.forward (aggregator.py:206) before
calling this module; here the im2row kernel does it, so `forward_features(x, normalized=False)`
takes raw [0,1] images by default and `normalized=True` is rejected (no un-fused path).
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _C, graphs, precision
from .blocks import Block, MemEffAttention, Mlp, Workspace, compensated_bias
from .patch_embed import PatchEmbed

KPAD = 640  # 3*14*14 = 588 taps padded to a multiple of the GEMM K-tile (64)


class DinoVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=True, ffn_bias=True, proj_bias=True, drop_path_rate=0.0,
                 drop_path_uniform=False, init_values=None, embed_layer=PatchEmbed, act_layer=nn.GELU,
                 block_fn=Block, ffn_layer="mlp", block_chunks=1, num_register_tokens=0,
                 interpolate_antialias=False, interpolate_offset=0.1, qk_norm=False):
        super().__init__()
        if ffn_layer != "mlp" or block_chunks not in (0, 1):
            raise NotImplementedError("only the ViT configuration used by IGGT (mlp FFN, unchunked) is built")
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        self.n_blocks = depth
        self.num_heads = num_heads
        self.patch_size = patch_size
        self.num_register_tokens = num_register_tokens
        self.interpolate_antialias = interpolate_antialias
        self.interpolate_offset = interpolate_offset
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                       embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))
        self.register_tokens = (nn.Parameter(torch.zeros(1, num_register_tokens, embed_dim))
                                if num_register_tokens else None)
        self.blocks = nn.ModuleList([
            block_fn(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                     proj_bias=proj_bias, ffn_bias=ffn_bias, norm_layer=norm_layer, act_layer=act_layer,
                     ffn_layer=Mlp, init_values=init_values, qk_norm=qk_norm)
            for _ in range(depth)])
        self.chunked_blocks = False
        self.norm = norm_layer(embed_dim)
        self.head = nn.Identity()
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        if self.register_tokens is not None:
            nn.init.normal_(self.register_tokens, std=1e-6)
        self._cache = {}
        self._ws = Workspace()

    # ---- host-side, input-independent tables (cached per (H, W, param version)) -------------------
    def _pos_tables(self, H, W):
        """(special [5, C] fp32 = [cls + pos_cls, registers], pos_patch [g2, C] fp32).
        Bicubic-antialias resample of the 37x37 table when (H, W) != (518, 518):
        reference vision_transformer.py:183-215 (`size=` branch, interpolate_offset = 0.0)."""
        key = (H, W, self.pos_embed._version, self.pos_embed.data_ptr(), self.cls_token._version,
               None if self.register_tokens is None else self.register_tokens._version)
        tables = self._cache.setdefault("pos_tables", {})
        if key not in tables:
            # one entry per (H, W, parameter version): alternating input shapes must not free each other's tables -- captured
            # graphs hold their addresses (graphs.py); the oldest entries go once 16 shapes have been seen
            while len(tables) >= 16:
                tables.pop(next(iter(tables)))
                graphs.buffers_changed()
            pe = self.pos_embed.detach().float()
            N = pe.shape[1] - 1
            gh, gw = H // self.patch_size, W // self.patch_size
            if gh * gw == N and H == W:
                patch_pe = pe[0, 1:]
            else:
                if self.interpolate_offset:
                    raise NotImplementedError("interpolate_offset != 0 is not used by IGGT (aggregator.py:149)")
                M = int(math.sqrt(N))
                assert N == M * M
                patch_pe = F.interpolate(pe[:, 1:].reshape(1, M, M, -1).permute(0, 3, 1, 2), mode="bicubic",
                                         antialias=self.interpolate_antialias, size=(gh, gw))
                patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(gh * gw, -1)
            special = [self.cls_token.detach().float()[0] + pe[0, :1]]
            if self.register_tokens is not None:
                special.append(self.register_tokens.detach().float()[0])
            tables[key] = (torch.cat(special, 0).contiguous(), patch_pe.contiguous())
        return tables[key]

    def _packed_patch_weight(self):
        w = self.patch_embed.proj.weight
        dt = precision.operand_dtype()
        key = (w.data_ptr(), w._version, dt, precision.mean_compensation())
        if self._cache.get("pw_key") != key:
            if "pw" in self._cache:
                graphs.buffers_changed()
            w2 = w.detach().reshape(w.shape[0], -1).float()
            wp = torch.zeros(w.shape[0], KPAD, dtype=dt, device=w.device)
            wp[:, : w2.shape[1]] = w2.to(dt)
            dwp = None
            if precision.mean_compensation():   # dW = W - round16(W) for the mean-input compensation (blocks.py)
                dwp = torch.zeros_like(wp)
                dwp[:, : w2.shape[1]] = (w2 - w2.to(dt).float()).to(dt)
            self._cache["pw_key"] = key
            self._cache["pw"] = (wp, self.patch_embed.proj.bias.detach().float().contiguous(), dwp)
        return self._cache["pw"]

    def _packed_patch_weight_x3(self):
        w = self.patch_embed.proj.weight
        key = (w.data_ptr(), w._version)
        if self._cache.get("pw3_key") != key:
            if "pw3" in self._cache:
                graphs.buffers_changed()
            w2 = w.detach().reshape(w.shape[0], -1).float()
            wpad = torch.zeros(w.shape[0], KPAD, dtype=torch.float32, device=w.device)
            wpad[:, : w2.shape[1]] = w2
            from .blocks import _x3_weight

            self._cache["pw3_key"] = key
            self._cache["pw3"] = _x3_weight(wpad)
        return self._cache["pw3"]

    # ---- fused forward ---------------------------------------------------------------------------
    def forward_tokens(self, images: torch.Tensor) -> torch.Tensor:
        """images: raw [0,1] fp32 [S,3,H,W] on the GPU -> pre-norm tokens x fp32 [S, 5+g2, C]."""
        if not images.is_cuda:
            raise _C.HipExtensionError("DinoVisionTransformer runs on HIP kernels only (no CPU fallback)")
        S, C_in, H, W = images.shape
        ps = self.patch_size
        assert ps == 14 and C_in == 3, "patch-embed kernel is built for 3x14x14 patches"
        assert H % ps == 0, f"Input image height {H} is not a multiple of patch height {ps}"
        assert W % ps == 0, f"Input image width {W} is not a multiple of patch width: {ps}"
        gh, gw = H // ps, W // ps
        g2 = gh * gw
        nsp = 1 + self.num_register_tokens
        P = nsp + g2
        D = self.embed_dim
        dev = images.device
        images = images.contiguous().float()
        special, patch_pe = self._pos_tables(H, W)
        wp, bias, dwp = self._packed_patch_weight()
        x = torch.empty(S, P, D, dtype=torch.float32, device=dev)
        if any(blk.packed()["x3"] for blk in self.blocks):
            # the patch embedding follows its consumers onto the x3 precision rung (precision.py): pixels and weights as fp16
            # hi + lo pairs, one GEMM over the concatenated K axis (layers/blocks.py _x3_weight)
            a3 = self._ws.get("im2row_x3", (S * g2, 3 * KPAD), torch.float16, dev)
            _C.im2row_patch14(images, a3, S, H, W, KPAD, split3=True)
            _C.gemm_h16(a3, self._packed_patch_weight_x3(), x.view(S * P, D), bias=bias, add_table=patch_pe, rows_in=g2,
                         rows_out=P, row_off=nsp)
        else:
            a = self._ws.get("im2row", (S * g2, KPAD), wp.dtype, dev)
            _C.im2row_patch14(images, a, S, H, W, KPAD)
            _C.gemm_h16(a, wp, x.view(S * P, D), bias=compensated_bias(self._ws, a, dwp, bias), add_table=patch_pe,
                         rows_in=g2, rows_out=P, row_off=nsp)
        _C.write_special_tokens(x, special, special, S, nsp, 0, False)
        x2d = x.view(S * P, D)
        for blk in self.blocks:
            blk.forward_inplace(x2d, self._ws, batch=S, tokens=P)
        return x

    def patch_tokens_into(self, images: torch.Tensor, dst: torch.Tensor, dst_row_off: int):
        """Final LayerNorm of the patch rows written straight into dst[s, dst_row_off + i, :]
        (the aggregator's token buffer) -- reference: `x_norm_patchtokens` (vision_transformer.py:274-278)."""
        x = self.forward_tokens(images)
        S, P, D = x.shape
        nsp = 1 + self.num_register_tokens
        g2 = P - nsp
        assert dst.shape[0] == S and dst.shape[2] == D and dst.stride(2) == 1
        assert dst.stride(0) % dst.stride(1) == 0
        w, b = self.norm.weight.detach().float(), self.norm.bias.detach().float()
        # rows (s, i) -> read x[s, nsp + i], write dst[s, dst_row_off + i]
        _C.layernorm(x, w, b, dst, self.norm.eps, rows=S * g2, rows_in=g2, rows_stride=P, row_off=nsp,
                     orows_stride=dst.stride(0) // dst.stride(1), orow_off=dst_row_off)
        return dst

    def forward_features(self, x, masks=None):
        if masks is not None:
            raise NotImplementedError("mask tokens are a training-time feature (out of scope)")
        S, _, H, W = x.shape
        nsp = 1 + self.num_register_tokens
        g2 = (H // self.patch_size) * (W // self.patch_size)
        pre = self.forward_tokens(x)
        normed = torch.empty_like(pre)
        w, b = self.norm.weight.detach().float(), self.norm.bias.detach().float()
        _C.layernorm(pre.view(-1, pre.shape[-1]), w, b, normed.view(-1, pre.shape[-1]), self.norm.eps)
        return {
            "x_norm_clstoken": normed[:, 0],
            "x_norm_regtokens": normed[:, 1:nsp],
            "x_norm_patchtokens": normed[:, nsp:nsp + g2],
            "x_prenorm": pre,
            "masks": masks,
        }

    def forward(self, *args, is_training=True, **kwargs):
        ret = self.forward_features(*args, **kwargs)
        return ret if is_training else self.head(ret["x_norm_clstoken"])


def vit_large(patch_size=16, num_register_tokens=0, **kwargs):
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention),
                                 num_register_tokens=num_register_tokens, **kwargs)


def _unsupported(name):
    def f(*a, **k):
        raise NotImplementedError(f"{name}: IGGT uses dinov2_vitl14_reg only (aggregator.py:59); head_dim-64 "
                                  "HIP kernels are built for ViT-L")
    return f


vit_small, vit_base, vit_giant2 = _unsupported("vit_small"), _unsupported("vit_base"), _unsupported("vit_giant2")
