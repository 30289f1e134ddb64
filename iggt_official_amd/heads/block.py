"""Attention blocks of the part head (reference iggt/heads/block.py:101-283).

The reference runs xformers' memory-efficient attention when available and an explicit
softmax(q k^T * scale) v otherwise (block.py:132-136, 225-229); both are the same function.  Here the
attention core is `F.scaled_dot_product_attention` on the GPU in fp32 (head dim 32: the head-dim-64
HIP flash kernel does not apply; DESIGN.md lists a d=32 variant as next); the q/k/v/proj Linear layers run on the
split-bf16 HIP GEMM (heads/tokenops.py).  qk_norm / rope are never
enabled by IGGT (part_head.py:74,83; window_sa.py:193) and are not built.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import tokenops as tk


class Attention(nn.Module):
    def __init__(self, dim, rope=None, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0, qk_norm=False):
        super().__init__()
        if rope is not None or qk_norm:
            raise NotImplementedError("rope / qk_norm are unused by IGGT's part head")
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.rope, self.qk_norm = None, False

    def forward(self, x, xpos=None):
        B, N, C = x.shape
        qkv = tk.linear(self.qkv, x).view(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], scale=self.scale)
        return tk.linear(self.proj, o.transpose(1, 2).reshape(B, N, C))


MemEffAttention = Attention


class CrossAttention(nn.Module):
    def __init__(self, dim, rope=None, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0,
                 use_xformers=False, qk_norm=False):
        super().__init__()
        if rope is not None or qk_norm:
            raise NotImplementedError("rope / qk_norm are unused by IGGT's part head")
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.projq = nn.Linear(dim, dim, bias=qkv_bias)
        self.projk = nn.Linear(dim, dim, bias=qkv_bias)
        self.projv = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.rope, self.qk_norm = None, False

    def forward(self, query, key, value, qpos=None, kpos=None):
        B, Nq, C = query.shape
        h, d = self.num_heads, self.head_dim
        q = tk.linear(self.projq, query).view(B, Nq, h, d).transpose(1, 2)
        k = tk.linear(self.projk, key).view(B, key.shape[1], h, d).transpose(1, 2)
        v = tk.linear(self.projv, value).view(B, value.shape[1], h, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, scale=self.scale)
        return tk.linear(self.proj, o.transpose(1, 2).reshape(B, Nq, C))


class MemEffCrossAttention(CrossAttention):
    def __init__(self, dim, rope=None, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0, qk_norm=False):
        super().__init__(dim, rope, num_heads, qkv_bias, attn_drop, proj_drop)
