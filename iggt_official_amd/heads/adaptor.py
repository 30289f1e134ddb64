"""Token -> multi-scale feature adaptors (reference iggt/heads/adaptor.py:9-226).

`SamProjector` ("part_adaptor" of IGGT, vggt.py:145-146) turns the four kept aggregator layers into
a 4-level pyramid res1..res4 (4g, 2g, g, ceil(g/2)) through transposed convs and `Projects`
(1x1 conv + BN + ReLU -> two 3x3 conv + BN with a skip -> 1x1 conv).  Everything runs NHWC on the HIP
kernels: the token stage is the shared LayerNorm + GEMM path of dpt_head.TokenProjector; every conv is
the implicit-GEMM MFMA kernel with the eval-mode BatchNorms folded into the weights at pack time;
ConvTranspose2d(k4,s2,p1) is four 2x2 parity convolutions, ConvTranspose2d(k2,s2) a GEMM + pixel shuffle.
The reference also evaluates a SAM-2 sine position encoding whose result is discarded by the caller
(adaptor.py:223, vggt.py:208): it is not computed here and `pos` is returned as an empty dict.
"""
import torch.nn as nn

from . import convops as co
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
        self._pk = co.PackCache()

    def _packs(self):
        ip, rc = self.input_proj, self.residual_conv
        tensors = [ip[0].weight, ip[1].weight, ip[1].bias, ip[1].running_mean, ip[1].running_var,
                   rc[0].weight, rc[1].weight, rc[1].bias, rc[1].running_mean, rc[1].running_var,
                   rc[3].weight, rc[4].weight, rc[4].bias, rc[4].running_mean, rc[4].running_var,
                   self.output_proj.weight, self.output_proj.bias]
        return self._pk.get(0, tensors, lambda: (co.pack_conv2d(ip[0], ip[1]), co.pack_conv2d(rc[0], rc[1]),
                                                 co.pack_conv2d(rc[3], rc[4]), co.pack_conv2d(self.output_proj)))

    def forward_nhwc(self, x):
        if self.training:
            raise RuntimeError("Projects folds BatchNorm running statistics: call model.eval()")
        p0, p1, p2, p3 = self._packs()
        x1 = co.run(p0, x, act=1)                 # relu(bn(conv1x1 x))
        y = co.run(p1, x1, act=1, prec=co.PART_PREC)     # relu(bn(conv3x3))
        y = co.run(p2, y, res=x1, prec=co.PART_PREC)     # bn(conv3x3) + skip
        return co.run(p3, y)


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
        self._pk = co.PackCache()

    def _run_layer(self, key, layer, x):
        """One layer of a resize stack on an NHWC map."""
        if isinstance(layer, nn.Identity):
            return x
        if isinstance(layer, Projects):
            return layer.forward_nhwc(x)
        if isinstance(layer, nn.Conv2d):
            return co.run(self._pk.get(key, (layer.weight, layer.bias), lambda: co.pack_conv2d(layer)), x)
        if isinstance(layer, nn.ConvTranspose2d):
            if layer.kernel_size == layer.stride and layer.padding == (0, 0):
                return co.run(self._pk.get(key, (layer.weight, layer.bias),
                                           lambda: co.pack_convT_kernel_eq_stride(layer)), x)
            return co.run_convT_k4s2p1(self._pk.get(key, (layer.weight, layer.bias),
                                                    lambda: co.pack_convT_k4s2p1(layer)), x)
        if isinstance(layer, nn.Sequential):
            for j, sub in enumerate(layer):
                x = self._run_layer((key, j), sub, x)
            return x
        raise TypeError(type(layer))

    def _pyramid(self, tokens_list, images, psi, s0=None, s1=None):
        """-> {key: NHWC fp32 map}"""
        _, S, _, H, W = images.shape
        s0 = 0 if s0 is None else s0
        s1 = S if s1 is None else s1
        gh, gw = H // self.patch_size, W // self.patch_size
        out = {}
        for i, (li, key) in enumerate(zip(self.intermediate_layer_idx, self.KEYS)):
            conv = self.projects[i]
            pos = pos_embed_map(conv.out_channels, gh, gw, W, H, tokens_list[li].device) if self.pos_embed else None
            x = self._tp(tokens_list[li], s0, s1, psi, gh, gw, self.norm, i, conv, pos)
            out[key] = self._run_layer(("rl", i), self.resize_layers[i], x)
        return out

    def forward(self, aggregated_tokens_list, images, patch_start_idx, frames_start_idx=None, frames_end_idx=None):
        pyr = self._pyramid(aggregated_tokens_list, images, patch_start_idx, frames_start_idx, frames_end_idx)
        return {k: v.permute(0, 3, 1, 2) for k, v in pyr.items()}   # reference layout: NCHW views


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
        """-> (dict res1..res4 of NCHW-shaped views [S,256,.,.] (channels-last memory), {})."""
        pyr = self._pyramid(aggregated_tokens_list, images, patch_start_idx, frames_start_idx, frames_end_idx)
        return {k: v.permute(0, 3, 1, 2) for k, v in pyr.items()}, {}
