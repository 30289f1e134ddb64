"""Token -> multi-scale feature adaptors (reference iggt/heads/adaptor.py:9-226).

`SamProjector` ("part_adaptor" of IGGT, vggt.py:145-146) turns the four kept aggregator layers into
a 4-level pyramid res1..res4 (4g, 2g, g, ceil(g/2)) through transposed convs and `Projects`
(1x1 conv + BN + ReLU -> two 3x3 conv + BN with a skip -> 1x1 conv).  The token stage
(LayerNorm + 1x1 projection) is the shared HIP path of dpt_head.TokenProjector; the conv stacks run
through PyTorch-ROCm in fp32 (BatchNorm in eval mode).  The reference also evaluates a SAM-2 sine
position encoding whose result is discarded by the caller (adaptor.py:223, vggt.py:208): it is not
computed here and `pos` is returned as an empty dict.
"""
from typing import List

import torch
import torch.nn as nn

from .dpt_head import TokenProjector
from .utils import pos_embed_map


class Projects(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.input_proj = nn.Sequential(nn.Conv2d(dim_in, dim_out, 1, 1, 0, bias=False), nn.BatchNorm2d(dim_out),
                                        nn.ReLU(inplace=True))
        self.residual_conv = nn.Sequential(nn.Conv2d(dim_out, dim_out, 3, 1, 1, bias=False),
                                           nn.BatchNorm2d(dim_out), nn.ReLU(inplace=True),
                                           nn.Conv2d(dim_out, dim_out, 3, 1, 1, bias=False),
                                           nn.BatchNorm2d(dim_out))
        self.output_proj = nn.Conv2d(dim_out, dim_out, 1)

    def forward(self, x):
        x = self.input_proj(x)
        return self.output_proj(self.residual_conv(x) + x)


class GeoProjector(nn.Module):
    KEYS = ["res2", "res3", "res4", "res5"]

    def __init__(self, dim_in, patch_size=14, pos_embed=False, intermediate_layer_idx=[4, 11, 17, 23],
                 out_channels=[256, 256, 256, 256]):
        super().__init__()
        self.out_channels = out_channels
        self.intermediate_layer_idx = intermediate_layer_idx
        self.patch_size = patch_size
        self.pos_embed = pos_embed
        self.norm = nn.LayerNorm(dim_in)
        self.projects = nn.ModuleList([nn.Conv2d(dim_in, oc, 1, 1, 0) for oc in out_channels])
        self.resize_layers = nn.ModuleList([
            nn.ConvTranspose2d(out_channels[0], out_channels[0], 4, 4, 0),
            nn.ConvTranspose2d(out_channels[1], out_channels[1], 2, 2, 0),
            nn.Identity(),
            nn.Conv2d(out_channels[3], out_channels[3], 3, 2, 1),
        ])
        self._tp = TokenProjector()

    def _pyramid(self, tokens_list, images, psi, s0=None, s1=None):
        _, S, _, H, W = images.shape
        s0 = 0 if s0 is None else s0
        s1 = S if s1 is None else s1
        gh, gw = H // self.patch_size, W // self.patch_size
        out = {}
        for i, (li, key) in enumerate(zip(self.intermediate_layer_idx, self.KEYS)):
            conv = self.projects[i]
            pos = pos_embed_map(conv.out_channels, gh, gw, W, H, tokens_list[li].device) if self.pos_embed else None
            x = self._tp(tokens_list[li], s0, s1, psi, gh, gw, self.norm, i, conv, pos)
            out[key] = self.resize_layers[i](x)
        return out

    def forward(self, aggregated_tokens_list, images, patch_start_idx, frames_start_idx=None, frames_end_idx=None):
        return self._pyramid(aggregated_tokens_list, images, patch_start_idx, frames_start_idx, frames_end_idx)


class SamProjector(GeoProjector):
    KEYS = ["res1", "res2", "res3", "res4"]

    def __init__(self, dim_in, patch_size=14, pos_embed=False, intermediate_layer_idx=[4, 11, 17, 23],
                 out_channels=[256, 256, 256, 256]):
        super().__init__(dim_in, patch_size, pos_embed, intermediate_layer_idx, out_channels)
        oc = out_channels
        self.resize_layers = nn.ModuleList([
            nn.Sequential(nn.ConvTranspose2d(oc[0], oc[0], 4, 2, 1), Projects(oc[0], oc[0]),
                          nn.ConvTranspose2d(oc[0], oc[0], 4, 2, 1), Projects(oc[0], oc[0])),
            nn.Sequential(nn.ConvTranspose2d(oc[1], oc[1], 2, 2, 0), Projects(oc[1], oc[1])),
            nn.Sequential(nn.Identity(), Projects(oc[2], oc[2])),
            nn.Sequential(nn.Conv2d(oc[3], oc[3], 3, 2, 1), Projects(oc[3], oc[3])),
        ])

    def forward(self, aggregated_tokens_list, images, patch_start_idx, frames_start_idx=None, frames_end_idx=None):
        out = self._pyramid(aggregated_tokens_list, images, patch_start_idx, frames_start_idx, frames_end_idx)
        return out, {}
