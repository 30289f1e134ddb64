"""PatchEmbed parameter holder (reference iggt/layers/patch_embed.py:25-88).

The 14x14/14 convolution runs as im2row (`iggt_im2row_patch14`, fused with the ImageNet
normalisation) + the bf16 MFMA GEMM with bias/pos-embed epilogue; see
models/aggregator.py `_patch_tokens`.  This module keeps the reference's parameter names
(`proj.weight [1024,3,14,14]`, `proj.bias`).
"""
import torch.nn as nn


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None,
                 flatten_embedding=True):
        super().__init__()
        hw = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        ps = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size, self.patch_size = hw, ps
        self.patches_resolution = (hw[0] // ps[0], hw[1] // ps[1])
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.flatten_embedding = flatten_embedding
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=ps, stride=ps)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        raise RuntimeError("PatchEmbed runs fused inside DinoVisionTransformer (HIP im2row + GEMM); "
                           "call the enclosing module")
