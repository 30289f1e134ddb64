"""Attention blocks of the part head (reference iggt/heads/block.py:101-283).

The reference runs xformers' memory-efficient attention when available and an explicit
softmax(q k^T * scale) v otherwise (block.py:132-136, 225-229); both are the same function.  Here the
attention core is the fp32 HIP kernel `iggt_attn_f32` (head dim 32, token-major operands read in place from the
projection outputs: no head transposes); the q/k/v/proj Linear layers run on the split-bf16 HIP GEMM
(heads/tokenops.py).  qk_norm / rope are never enabled by IGGT (part_head.py:74,83; window_sa.py:193) and are not built.
"""
import torch
import torch.nn as nn

from .. import _C, profiling
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
        qkv = tk.linear(self.qkv, x).view(B * N, 3 * C)
        o = torch.empty(B, N, C, dtype=torch.float32, device=x.device)
        _C.attn_f32(qkv, qkv[:, C:], qkv[:, 2 * C:], o, B, self.num_heads, N, N, self.head_dim,
                    N * 3 * C, 3 * C, N * 3 * C, 3 * C, N * 3 * C, 3 * C, N * C, C, self.scale)
        return tk.linear(self.proj, o)


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
        Nk = key.shape[1]
        q = tk.linear(self.projq, query)     # [B, Nq, C] fp32, head h in columns [h d, (h + 1) d)
        k = tk.linear(self.projk, key)
        v = tk.linear(self.projv, value)
        o = torch.empty(B, Nq, C, dtype=torch.float32, device=q.device)
        with profiling.region("cross_attn", ("fp32 MFMA", 4.0 * B * Nq * Nk * C)):     # (kind, algorithmic FLOPs)
            _C.attn_f32(q, k, v, o, B, self.num_heads, Nq, Nk, self.head_dim, Nq * C, C, Nk * C, C, Nk * C, C, Nq * C, C,
                        self.scale)
        return tk.linear(self.proj, o)


class MemEffCrossAttention(CrossAttention):
    def __init__(self, dim, rope=None, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0, qk_norm=False):
        super().__init__(dim, rope, num_heads, qkv_bias, attn_drop, proj_drop)
