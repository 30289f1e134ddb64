"""Instance-feature ("part") head -- reference iggt/heads/part_head.py:14-243.

A DPT-style fusion over the SamProjector pyramid (res1..res4, 256 ch each) that is conditioned on
the point head's fusion features (out2 @4g, out3 @2g, out4 @g):
  refinenet4 -> cross_attention_2(q = part tokens @g, k = v = point out4 @g)   (part_head.py:168-173)
  -> refinenet3 -> refinenet2 -> SwinCA window cross-attention with point out2 @4g (188-197)
  -> refinenet1 -> output_conv1 (128 ch @8g) -> SwinSA window self-attention (222-225)
  -> bilinear(align_corners) to HxW -> conv3x3 -> ReLU -> conv1x1 -> [B,S,8,H,W], NO activation
  (240-243; the ctor's activation="norm" is never applied).
Reference quirks kept: `cross_attention_1` is evaluated by the reference but its result is dead
(part_head.py:178-185, appendix D.3) -- its parameters exist for checkpoint loading, the compute is
skipped; PartHead inherits DPTHead.__init__ so unused `norm/projects/resize_layers` parameters
exist as well (appendix D.6).
"""
from typing import List

import torch
import torch.nn as nn

from .block import MemEffCrossAttention
from .dpt_head import DPTHead, _make_fusion_block, _make_scratch, custom_interpolate
from .window_sa import SwinCA, SwinSA


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
        """aggregated_tokens_list: the 4 SamProjector maps [S, 256, ., .]; point_feature: (out2, out3, out4)."""
        B, S, _, H, W = images.shape
        if H % 28 or W % 28:
            raise ValueError(f"part_feat needs H and W to be multiples of 28 (got {H}x{W}): the 4g / 8g maps "
                             "must tile into 8x8 windows (reference window_sa.py:73,411)")
        chunk = S if (frames_chunk_size is None or frames_chunk_size >= S) else frames_chunk_size
        outs = []
        for s0 in range(0, S, chunk):
            s1 = min(s0 + chunk, S)
            maps = [m[s0:s1] for m in aggregated_tokens_list]
            pf = None if point_feature is None else [p[s0:s1] for p in point_feature]
            outs.append(self._part_impl(maps, H, W, pf))
        out = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        return out.view(B, S, *out.shape[1:])

    def _fuse(self, features, point_feat):
        l1, l2, l3, l4 = features
        sc = self.scratch
        r1, r2, r3, r4 = sc.layer1_rn(l1), sc.layer2_rn(l2), sc.layer3_rn(l3), sc.layer4_rn(l4)
        out = sc.refinenet4(r4, size=r3.shape[2:])
        if point_feat is not None:
            q = out.flatten(2).transpose(1, 2)
            kv = point_feat[2].flatten(2).transpose(1, 2)
            out4 = self.cross_attention_2(q, kv, kv).transpose(1, 2).reshape(out.shape)
        else:
            out4 = out
        out = sc.refinenet3(out4, r3, size=r2.shape[2:])
        out = sc.refinenet2(out, r2, size=r1.shape[2:])
        if point_feat is not None:
            out2 = self.window_cross_attention(out.permute(0, 2, 3, 1), point_feat[0].permute(0, 2, 3, 1),
                                               point_feat[0].permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        else:
            out2 = out
        out = sc.refinenet1(out2, r1)
        return sc.output_conv1(out)

    def _part_impl(self, maps, H, W, point_feat):
        gh, gw = H // self.patch_size, W // self.patch_size
        out = self._fuse(maps, point_feat)
        out = self.window_self_atten(out.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2).contiguous()
        out = custom_interpolate(out, (int(gh * self.patch_size / self.down_ratio),
                                       int(gw * self.patch_size / self.down_ratio)))
        return self.scratch.output_conv2(out)
