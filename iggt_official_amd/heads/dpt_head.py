"""DPT dense-prediction head (depth / point maps) -- reference iggt/heads/dpt_head.py:21-509.

State-dict layout is the reference's (`norm`, `projects.{0..3}`, `resize_layers.{0,1,3}`,
`scratch.layer{1..4}_rn`, `scratch.refinenet{1..4}.{out_conv,resConfUnit{1,2}.conv{1,2}}`,
`scratch.output_conv1`, `scratch.output_conv2.{0,2}`).

Execution on the MI355X -- everything NHWC fp32 in HBM, all GEMM-shaped work on MFMA:
  * token stage: LayerNorm(2048) over the [frame|global] halves with the 5 special tokens skipped by row
    remap (csrc/elementwise.hip), then the 1x1 `projects` conv as a bf16 MFMA GEMM whose epilogue adds the
    bias and the cached UV sin/cos position map (csrc/gemm_bf16.hip) -> NHWC directly;
  * every Conv2d / ConvTranspose2d runs on the implicit-GEMM kernel (csrc/conv_igemm.hip) with split-bf16
    operands (fp32-grade, the reference runs the heads in fp32: vggt.py:189): transposed convs with
    kernel == stride are 1x1 GEMMs with a pixel-shuffle epilogue; ResidualConvUnit is two launches
    (ReLU-on-load -> conv -> ReLU, then conv + rectified-input skip [+ the fusion add]);
  * bilinear align_corners resizes (+ the full-resolution position map) are one HBM-bound kernel;
  * tail: conv3x3(128->32)+ReLU on MFMA; the 32->{4,2} 1x1 and the exp/expm1 activations are a few
    PyTorch-ROCm elementwise ops on [S,H,W,4].
Frames are independent in every head op, so `frames_chunk_size` only bounds
memory; results do not depend on it (the reference's own chunked branch is broken for S > 12, App. D.1).
"""
import os
from typing import List

import torch
import torch.nn as nn

from .. import _C, graphs, precision
from ..layers.blocks import Workspace, compensated_bias
from . import convops as co
from .utils import pos_embed_map, pos_embed_rows


def custom_interpolate(x, size=None, scale_factor=None, mode="bilinear", align_corners=True):
    """Reference dpt_head.py:484-509 (NCHW in, NCHW out) on the HIP resize kernel (csrc/dpt_tail.hip
    bilinear_ac_nhwc_kernel).  The reference only ever calls it as bilinear / align_corners=True; anything else raises --
    there is no torch fallback in the product.  Any channel count (the kernel moves 8 channels per thread: other counts are
    zero-padded for the call); returns a contiguous fp32 NCHW tensor without autograd history (inference path)."""
    if mode != "bilinear" or not align_corners:
        raise _C.HipExtensionError("custom_interpolate: only bilinear, align_corners=True is built (the reference's use)")
    if x.dim() != 4 or not x.is_cuda:
        raise _C.HipExtensionError("custom_interpolate expects a [N, C, H, W] tensor on the GPU (no CPU fallback)")
    if size is None:
        if scale_factor is None:
            raise ValueError("custom_interpolate: either size or scale_factor must be given")
        size = (int(x.shape[-2] * scale_factor), int(x.shape[-1] * scale_factor))
    C = x.shape[1]
    nhwc = x.detach().float().permute(0, 2, 3, 1)
    if C % 8:
        nhwc = torch.nn.functional.pad(nhwc, (0, 8 - C % 8))
    out = co.resize(nhwc.contiguous(), tuple(int(v) for v in size))
    return out[..., :C].permute(0, 3, 1, 2).contiguous()


# FeatureFusionBlock: 1 x 1 out_conv before the upsampling instead of after it (exact identity, see forward_nhwc).
# IGGT_OUT_CONV_FIRST=0 restores the reference's order.
OUT_CONV_FIRST = os.environ.get("IGGT_OUT_CONV_FIRST", "1") != "0"


class ResidualConvUnit(nn.Module):
    """conv2(relu(conv1(relu(x)))) + relu(x): the skip adds the *rectified* input because the
    reference's activation is nn.ReLU(inplace=True) (dpt_head.py:327,401,411)."""

    def __init__(self, features, activation=None, bn=False, groups=1):
        super().__init__()
        self.bn, self.groups = bn, groups
        self.conv1 = nn.Conv2d(features, features, 3, 1, 1, bias=True, groups=groups)
        self.conv2 = nn.Conv2d(features, features, 3, 1, 1, bias=True, groups=groups)
        self.norm1 = None
        self.norm2 = None
        self._pk = co.PackCache()

    def forward_nhwc(self, x, extra=None, prec=None):
        """x NHWC -> conv2(relu(conv1(relu x))) + relu(x) (+ extra).  prec: operand precision of the two convolutions
        (None = convops.PREC; the DPT depth / point heads pass convops.DPT_PREC)."""
        p1 = self._pk.get(1, (self.conv1.weight, self.conv1.bias), lambda: co.pack_conv2d(self.conv1))
        p2 = self._pk.get(2, (self.conv2.weight, self.conv2.bias), lambda: co.pack_conv2d(self.conv2))
        t = co.run(p1, x, relu_in=True, act=1, prec=prec)
        return co.run(p2, t, res=x, relu_res=True, res2=extra, prec=prec)


class FeatureFusionBlock(nn.Module):
    def __init__(self, features, activation=None, deconv=False, bn=False, expand=False, align_corners=True,
                 size=None, has_residual=True, groups=1):
        super().__init__()
        self.align_corners = align_corners
        self.out_conv = nn.Conv2d(features, features // 2 if expand else features, 1, 1, 0, bias=True,
                                  groups=groups)
        if has_residual:
            self.resConfUnit1 = ResidualConvUnit(features, activation, bn, groups=groups)
        self.has_residual = has_residual
        self.resConfUnit2 = ResidualConvUnit(features, activation, bn, groups=groups)
        self.size = size
        self._pk = co.PackCache()

    def forward_nhwc(self, x0, x1=None, size=None, prec=None):
        """reference FeatureFusionBlock.forward (dpt_head.py:455-481) on NHWC tensors."""
        y = x0
        if self.has_residual:
            y = self.resConfUnit1.forward_nhwc(x1, extra=x0, prec=prec)      # x0 + rcu1(x1), the add is fused
        y = self.resConfUnit2.forward_nhwc(y, prec=prec)
        if size is None:
            size = self.size if self.size is not None else (2 * y.shape[1], 2 * y.shape[2])
        pc = self._pk.get(0, (self.out_conv.weight, self.out_conv.bias), lambda: co.pack_conv2d(self.out_conv))
        if OUT_CONV_FIRST and size[0] * size[1] > y.shape[1] * y.shape[2]:
            # The reference interpolates, then applies the 1 x 1 out_conv (dpt_head.py:471-479).  Both are linear maps that act
            # on different axes (pixels / channels) and the interpolation weights of a pixel sum to 1 (bias), so they commute
            # exactly; run the convolution on the SMALL map -- a quarter of the pixels (refinenet1: 2.1 -> 0.55 ms per head at 32
            # views).  Only the fp32 rounding order differs (~1e-7, tests/test_heads_gpu.py compares both orders).
            return co.resize(co.run(pc, y, prec=prec), tuple(size))
        y = co.resize(y, tuple(size))
        return co.run(pc, y, prec=prec)


def _make_fusion_block(features, size=None, has_residual=True, groups=1):
    return FeatureFusionBlock(features, None, deconv=False, bn=False, expand=False, align_corners=True,
                              size=size, has_residual=has_residual, groups=groups)


def _make_scratch(in_shape, out_shape, groups=1, expand=False):
    scratch = nn.Module()
    outs = [out_shape * (2 ** i if expand else 1) for i in range(4)]
    for i, cin in enumerate(in_shape[:4]):
        setattr(scratch, f"layer{i + 1}_rn", nn.Conv2d(cin, outs[i], 3, 1, 1, bias=False, groups=groups))
    return scratch


# Frames per head pass.  The reference defaults to 8 to bound memory on 24-80 GB GPUs (dpt_head.py:133); with 288 GB
# of HBM 32 frames fit easily (largest maps: 32 x 518^2 x 128 fp32 = 4.4 GB) and the larger launches fill the chip
# better: 70.0 / 71.6 / 72.2 views/s at 8 / 16 / 32 (32 views @ 518^2).  Results do not depend on it.
FRAMES_CHUNK = 32
# One fused kernel for upsample -> conv3x3 -> ReLU -> conv1x1 -> activations (IGGT_FUSED_TAIL=0: the separate kernels)
FUSED_TAIL = os.environ.get("IGGT_FUSED_TAIL", "1") != "0"


class TokenProjector:
    """LayerNorm(2C) + 1x1 conv on patch tokens, in HIP.  Shared by DPTHead and the adaptors."""

    def __init__(self):
        self.ws = Workspace()
        self._pk = {}

    def packed(self, idx, conv: nn.Conv2d):
        w = conv.weight
        dt = precision.operand_dtype()
        key = (w.data_ptr(), w._version, dt, precision.mean_compensation())
        if self._pk.get(idx, (None,))[0] != key:
            if idx in self._pk:
                graphs.buffers_changed()    # old pack freed: captured graphs hold its address
            w2 = w.detach().reshape(w.shape[0], -1).float()
            dw = (w2 - w2.to(dt).float()).to(dt).contiguous() if precision.mean_compensation() else None
            self._pk[idx] = (key, w2.to(dt).contiguous(), conv.bias.detach().float().contiguous(), dw)
        return self._pk[idx][1:]

    def __call__(self, tokens, s0, s1, psi, gh, gw, norm: nn.LayerNorm, idx, conv: nn.Conv2d, pos_table=None):
        """tokens [1, S_all, P, 2C] fp32 -> NHWC [S, gh, gw, oc] fp32."""
        if not tokens.is_cuda:
            raise _C.HipExtensionError("head token projection runs on HIP kernels only")
        _, S_all, P, C2 = tokens.shape
        S, g2 = s1 - s0, gh * gw
        t = tokens[0, s0:s1]
        assert t.is_contiguous() and P == psi + g2
        w, b, dw = self.packed(idx, conv)
        xn = self.ws.get("xn", (S * g2, C2), w.dtype, tokens.device)
        _C.layernorm(t.reshape(S * P, C2), norm.weight.detach().float(), norm.bias.detach().float(), xn,
                     norm.eps, rows=S * g2, rows_in=g2, rows_stride=P, row_off=psi)
        out = torch.empty(S * g2, w.shape[0], dtype=torch.float32, device=tokens.device)
        _C.gemm_h16(xn, w, out, bias=compensated_bias(self.ws, xn, dw, b), add_table=pos_table, rows_in=g2, rows_out=g2,
                    row_off=0)
        return out.view(S, gh, gw, w.shape[0])


class DPTHead(nn.Module):
    def __init__(self, dim_in, patch_size=14, output_dim=4, activation="inv_log", conf_activation="expp1",
                 features=256, out_channels=[256, 512, 1024, 1024], intermediate_layer_idx=[4, 11, 17, 23],
                 pos_embed=True, use_point_feat=False, down_ratio=1, for_tracker=False):
        super().__init__()
        self.patch_size = patch_size
        self.activation = activation
        self.conf_activation = conf_activation
        self.pos_embed = pos_embed
        self.for_tracker = for_tracker
        self.use_point_feat = use_point_feat
        self.down_ratio = down_ratio
        self.intermediate_layer_idx = intermediate_layer_idx
        self.norm = nn.LayerNorm(dim_in)
        self.projects = nn.ModuleList([nn.Conv2d(dim_in, oc, 1, 1, 0) for oc in out_channels])
        self.resize_layers = nn.ModuleList([
            nn.ConvTranspose2d(out_channels[0], out_channels[0], 4, 4, 0),
            nn.ConvTranspose2d(out_channels[1], out_channels[1], 2, 2, 0),
            nn.Identity(),
            nn.Conv2d(out_channels[3], out_channels[3], 3, 2, 1),
        ])
        self.scratch = _make_scratch(out_channels, features, expand=False)
        self.scratch.stem_transpose = None
        self.scratch.refinenet1 = _make_fusion_block(features)
        self.scratch.refinenet2 = _make_fusion_block(features)
        self.scratch.refinenet3 = _make_fusion_block(features)
        self.scratch.refinenet4 = _make_fusion_block(features, has_residual=False)
        h1, h2 = features, 32
        if for_tracker:
            self.scratch.output_conv1 = nn.Conv2d(h1, h1, 3, 1, 1)
        else:
            self.scratch.output_conv1 = nn.Conv2d(h1, h1 // 2, 3, 1, 1)
            self.scratch.output_conv2 = nn.Sequential(nn.Conv2d(h1 // 2, h2, 3, 1, 1), nn.ReLU(inplace=True),
                                                      nn.Conv2d(h2, output_dim, 1, 1, 0))
        self._tp = TokenProjector()
        self._pk = co.PackCache()

    # ------------------------------------------------------------------------------------------
    def _conv(self, key, conv):
        return self._pk.get(key, (conv.weight, conv.bias), lambda: co.pack_conv2d(conv))

    def _prec(self):
        """Operand precision of this head's convolutions: convops.DPT_PREC for the depth / point heads (measured per layer,
        profiles/r03_conv_precision.txt); the tracker's feature extractor stays at convops.PREC."""
        return None if self.for_tracker else co.DPT_PREC

    def forward(self, aggregated_tokens_list, images, patch_start_idx, frames_chunk_size=FRAMES_CHUNK):
        """Returns (preds [1,S,H,W,c], conf [1,S,H,W][, (out2,out3,out4) NHWC fusion features])."""
        B, S, _, H, W = images.shape
        if B != 1:
            raise NotImplementedError("heads run one scene (B=1) at a time, as demo.py does")
        chunk = S if (frames_chunk_size is None or frames_chunk_size >= S) else frames_chunk_size
        assert chunk > 0
        parts = [self._forward_impl(aggregated_tokens_list, H, W, patch_start_idx, s0, min(s0 + chunk, S))
                 for s0 in range(0, S, chunk)]
        if len(parts) == 1:
            return parts[0]
        if self.for_tracker:
            return torch.cat(parts, dim=1)
        merged = [torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 1)]
        if self.use_point_feat:
            merged.append(tuple(torch.cat([p[2][i] for p in parts], 0) for i in range(3)))
        return tuple(merged)

    def features_nhwc(self, aggregated_tokens_list, images, patch_start_idx, frames_chunk_size=FRAMES_CHUNK):
        """for_tracker heads: the feature map as the kernels produce it, NHWC [S, H / down_ratio, W / down_ratio, C]
        (`forward` returns the reference's [1, S, C, h, w] view of the same memory)."""
        assert self.for_tracker
        out = self.forward(aggregated_tokens_list, images, patch_start_idx, frames_chunk_size)
        return out[0].permute(0, 2, 3, 1).contiguous()     # no copy: forward() permuted a contiguous NHWC tensor

    def _token_maps(self, tokens_list, psi, s0, s1, H, W):
        """projects -> (+pos) -> resize_layers, NHWC (reference dpt_head.py:225-244)."""
        gh, gw = H // self.patch_size, W // self.patch_size
        maps = []
        for i, li in enumerate(self.intermediate_layer_idx):
            conv = self.projects[i]
            pos = pos_embed_map(conv.out_channels, gh, gw, W, H, tokens_list[li].device) if self.pos_embed else None
            x = self._tp(tokens_list[li], s0, s1, psi, gh, gw, self.norm, i, conv, pos)
            rl = self.resize_layers[i]
            if isinstance(rl, nn.ConvTranspose2d):
                pc = self._pk.get(("rl", i), (rl.weight, rl.bias), lambda rl=rl: co.pack_convT_kernel_eq_stride(rl))
                x = co.run(pc, x, prec=self._prec())
            elif isinstance(rl, nn.Conv2d):
                x = co.run(self._conv(("rl", i), rl), x, prec=self._prec())
            maps.append(x)
        return maps

    def _forward_impl(self, tokens_list, H, W, psi, s0, s1):
        S = s1 - s0
        gh, gw = H // self.patch_size, W // self.patch_size
        maps = self._token_maps(tokens_list, psi, s0, s1, H, W)
        out, side = self.scratch_forward(maps)
        size = (int(gh * self.patch_size / self.down_ratio), int(gw * self.patch_size / self.down_ratio))
        c2 = None if self.for_tracker else self.scratch.output_conv2
        fused = (not self.for_tracker and FUSED_TAIL and out.shape[3] == 128 and c2[0].in_channels == 128
                 and c2[0].out_channels == 32 and c2[0].kernel_size == (3, 3) and c2[0].padding == (1, 1)
                 and c2[2].in_channels == 32 and 2 <= c2[2].out_channels <= 8
                 and self.activation in ("linear", "exp", "relu", "inv_log", "sigmoid")
                 and self.conf_activation in _C.CONF_ACT)
        if fused:
            # upsample + position map + conv3x3 + ReLU + conv1x1 + activate_head in ONE kernel (csrc/dpt_tail.hip):
            # the full-resolution 128- and 32-channel maps never exist in HBM
            xr, yr = pos_embed_rows(128, size[0], size[1], W, H, out.device) if self.pos_embed else (None, None)
            pc = self._conv("oc2_0", c2[0])
            b1 = pc.bias if pc.bias is not None else torch.zeros(32, dtype=torch.float32, device=out.device)
            preds, conf = _C.dpt_tail(out.contiguous(), size, xr, yr, pc.w_hi, pc.w_lo, b1,
                                      c2[2].weight.detach().float().reshape(c2[2].out_channels, 32).contiguous(),
                                      c2[2].bias.detach().float().contiguous(), self.activation, self.conf_activation)
            preds = preds.reshape(1, S, *preds.shape[1:])
            conf = conf.reshape(1, S, *conf.shape[1:])
            return (preds, conf, side) if self.use_point_feat else (preds, conf)
        if self.pos_embed:
            xr, yr = pos_embed_rows(out.shape[3], size[0], size[1], W, H, out.device)
            out = co.resize(out, size, xr, yr)      # bilinear + position map in one pass
        else:
            out = co.resize(out, size)
        if self.for_tracker:
            return out.permute(0, 3, 1, 2)[None]
        out = co.run(self._conv("oc2_0", c2[0]), out, act=1, prec=self._prec())   # conv3x3 128->32 + ReLU
        if (c2[2].in_channels == 32 and 2 <= c2[2].out_channels <= 8 and self.activation in _C.HEAD_ACT
                and self.conf_activation in _C.CONF_ACT):
            # 1x1 conv 32 -> 4|2 + activate_head in one HBM pass (csrc/elementwise.hip head_tail_kernel)
            preds, conf = _C.head_tail(out, c2[2].weight.detach().float().reshape(c2[2].out_channels, 32).contiguous(),
                                       c2[2].bias.detach().float().contiguous(), self.activation,
                                       self.conf_activation)
        else:
            raise _C.HipExtensionError(
                f"DPTHead tail: the fused 1x1 convolution + activate_head kernel covers 32 -> 2..8 channels with activation in "
                f"{sorted(_C.HEAD_ACT)} / {sorted(_C.CONF_ACT)}; got {c2[2].in_channels} -> {c2[2].out_channels}, "
                f"{self.activation!r} / {self.conf_activation!r} (no torch fallback)")
        preds = preds.reshape(1, S, *preds.shape[1:])
        conf = conf.reshape(1, S, *conf.shape[1:])
        return (preds, conf, side) if self.use_point_feat else (preds, conf)

    def scratch_forward(self, features: List[torch.Tensor]):
        """reference dpt_head.py:286-316 on NHWC maps -> (output_conv1 map, (out2, out3, out4))."""
        sc, prec = self.scratch, self._prec()
        r = [co.run(self._conv(("rn", i), getattr(sc, f"layer{i + 1}_rn")), features[i], prec=prec) for i in range(4)]
        out4 = sc.refinenet4.forward_nhwc(r[3], size=r[2].shape[1:3], prec=prec)
        out3 = sc.refinenet3.forward_nhwc(out4, r[2], size=r[1].shape[1:3], prec=prec)
        out2 = sc.refinenet2.forward_nhwc(out3, r[1], size=r[0].shape[1:3], prec=prec)
        out1 = sc.refinenet1.forward_nhwc(out2, r[0], prec=prec)
        return co.run(self._conv("oc1", sc.output_conv1), out1, prec=prec), (out2, out3, out4)
