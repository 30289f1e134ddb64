"""Instance-feature ("part") head -- reference iggt/heads/part_head.py:14-243.

A DPT-style fusion over the SamProjector pyramid (res1..res4, 256 ch each) that is conditioned on
the point head's fusion features (out2 @4g, out3 @2g, out4 @g):
  refinenet4 -> cross_attention_2(q = part tokens @g, k = v = point out4 @g)   (part_head.py:168-173)
  -> refinenet3 -> refinenet2 -> SwinCA window cross-attention with point out2 @4g (188-197)
  -> refinenet1 -> output_conv1 (128 ch @8g) -> SwinSA window self-attention (222-225)
  -> bilinear(align_corners) to HxW -> conv3x3 -> ReLU -> conv1x1 -> [B,S,8,H,W], NO activation
  (240-243; the ctor's activation="norm" is never applied).
All maps are NHWC fp32; convolutions / resizes run on the HIP kernels (conv_igemm.hip), the token cross-attention core
(8 heads x 32) on the fp32 HIP attention kernel (csrc/smallops.hip), the last 1x1 convolution on a fused NHWC -> NCHW kernel.
Reference quirks kept: `cross_attention_1` is evaluated by the reference but its result is dead
(part_head.py:178-185, appendix D.3) -- its parameters exist for checkpoint loading, the compute is
skipped; PartHead inherits DPTHead.__init__ so unused `norm/projects/resize_layers` parameters
exist as well (appendix D.6).
"""
from typing import List

import torch
import torch.nn as nn

from .. import _C
from . import convops as co
from .block import MemEffCrossAttention
from .dpt_head import FUSED_TAIL, DPTHead, _make_fusion_block, _make_scratch
from .window_sa import SwinCA, SwinSA


def _nhwc(t):
    """Accept NCHW-shaped tensors (reference layout; channels-last views are free) -> contiguous NHWC."""
    return t.permute(0, 2, 3, 1).contiguous()


class PartHead(DPTHead):
    def __init__(self, dim_in, patch_size=14, output_dim=4, activation="relu", features=256,
                 out_channels=[256, 256, 256, 256], intermediate_layer_idx=[4, 11, 17, 23], pos_embed=True,
                 feature_only=False, down_ratio=1, for_tracker=False):
        super().__init__(dim_in=dim_in)
        self.for_tracker = for_tracker
        self.patch_size = patch_size
        self.activation = activation
        self.pos_embed = pos_embed
        self.feature_only = feature_only
        self.down_ratio = down_ratio
        self.intermediate_layer_idx = intermediate_layer_idx
        h1, h2 = features, 32
        self.scratch = _make_scratch(out_channels, features, expand=False)
        self.scratch.stem_transpose = None
        self.scratch.refinenet1 = _make_fusion_block(features)
        self.scratch.refinenet2 = _make_fusion_block(features)
        self.scratch.refinenet3 = _make_fusion_block(features)
        self.scratch.refinenet4 = _make_fusion_block(features, has_residual=False)
        self.scratch.output_conv1 = nn.Conv2d(h1, h1 // 2, 3, 1, 1)
        self.scratch.output_conv2 = nn.Sequential(nn.Conv2d(h1 // 2, h2, 3, 1, 1), nn.ReLU(inplace=True),
                                                  nn.Conv2d(h2, output_dim, 1, 1, 0))
        self.cross_attention_1 = MemEffCrossAttention(dim=h1, num_heads=8, qkv_bias=True)  # dead in forward
        self.cross_attention_2 = MemEffCrossAttention(dim=h1, num_heads=8, qkv_bias=True)
        self.window_self_atten = SwinSA(img_size=512, out_chans=h1 // 2, embed_dim=h1 // 2, num_heads=4,
                                        window_size=8)
        self.window_cross_attention = SwinCA(img_size=128, out_chans=h1, embed_dim=h1, num_heads=4, window_size=8)

    def forward(self, aggregated_tokens_list: List[torch.Tensor], images, patch_start_idx, frames_chunk_size=8,
                point_feature=None):
        """aggregated_tokens_list: the 4 SamProjector maps [S, 256, ., .] (NCHW-shaped);
        point_feature: (out2, out3, out4) NHWC maps of the point head."""
        B, S, _, H, W = images.shape
        if H % 28 or W % 28:
            raise ValueError(f"part_feat needs H and W to be multiples of 28 (got {H}x{W}): the 4g / 8g maps "
                             "must tile into 8x8 windows (reference window_sa.py:73,411)")
        chunk = S if (frames_chunk_size is None or frames_chunk_size >= S) else frames_chunk_size
        outs = []
        for s0 in range(0, S, chunk):
            s1 = min(s0 + chunk, S)
            maps = [_nhwc(m[s0:s1]) for m in aggregated_tokens_list]
            pf = None if point_feature is None else [p[s0:s1].contiguous() for p in point_feature]
            outs.append(self._part_impl(maps, H, W, pf))
        out = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        return out.view(B, S, *out.shape[1:])

    def _fuse(self, features, point_feat):
        sc = self.scratch
        pr = co.PART_PREC
        r = [co.run(self._conv(("rn", i), getattr(sc, f"layer{i + 1}_rn")), features[i], prec=pr) for i in range(4)]
        out = sc.refinenet4.forward_nhwc(r[3], size=r[2].shape[1:3], prec=pr)
        if point_feat is not None:
            n, h, w, c = out.shape
            kv = point_feat[2].reshape(n, -1, c)
            out = self.cross_attention_2(out.reshape(n, h * w, c), kv, kv).reshape(n, h, w, c)
        out = sc.refinenet3.forward_nhwc(out, r[2], size=r[1].shape[1:3], prec=pr)
        out = sc.refinenet2.forward_nhwc(out, r[1], size=r[0].shape[1:3], prec=pr)
        if point_feat is not None:
            out = self.window_cross_attention(out, point_feat[0], point_feat[0])
        out = sc.refinenet1.forward_nhwc(out, r[0], prec=pr)
        return co.run(self._conv("oc1", sc.output_conv1), out, prec=pr)

    def _part_impl(self, maps, H, W, point_feat):
        gh, gw = H // self.patch_size, W // self.patch_size
        out = self.window_self_atten(self._fuse(maps, point_feat))
        size = (int(gh * self.patch_size / self.down_ratio), int(gw * self.patch_size / self.down_ratio))
        c2 = self.scratch.output_conv2
        if (FUSED_TAIL and out.shape[3] == 128 and c2[0].in_channels == 128 and c2[0].out_channels == 32
                and c2[0].kernel_size == (3, 3) and c2[0].padding == (1, 1) and c2[2].in_channels == 32
                and 2 <= c2[2].out_channels <= 8):
            # round 6: bilinear upsample + conv3x3 128 -> 32 + ReLU + conv1x1 -> the NCHW output in ONE kernel (csrc/dpt_tail.hip,
            # the DPT heads' fused tail without position map and activation): the full-resolution 128- and 32-channel maps
            # (4.6 + 1.2 GB at 32 x 532^2) never exist in HBM
            pc = self._conv("oc2_0", c2[0])
            b1 = pc.bias if pc.bias is not None else torch.zeros(32, dtype=torch.float32, device=out.device)
            return _C.dpt_tail(out.contiguous(), size, None, None, pc.w_hi, pc.w_lo, b1,
                               c2[2].weight.detach().float().reshape(c2[2].out_channels, 32).contiguous(),
                               c2[2].bias.detach().float().contiguous(), "linear", "expp1", nchw=True)
        out = co.resize(out, size)
        out = co.run(self._conv("oc2_0", c2[0]), out, act=1)
        # last 1x1 conv, NHWC in -> the reference's NCHW [S,8,H,W] out in one pass (no activation: part_head.py:240-243)
        return _C.conv1x1_c32_nchw(out, c2[2].weight.detach(), c2[2].bias.detach())
