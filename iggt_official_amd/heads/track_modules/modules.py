"""Mlp / AttnBlock / CrossAttnBlock of the tracker's update transformer -- reference iggt/heads/track_modules/modules.py.

Parameter holders with the reference's state-dict names (nn.MultiheadAttention keeps `in_proj_weight`, `in_proj_bias`,
`out_proj.{weight,bias}`); the arithmetic runs on the fp32 HIP kernels: `iggt_layernorm_rows_f32`, `iggt_linear_f32`
(exact fp32 MFMA) and `iggt_attn_f32`.  The 8 heads are 48 wide; the attention kernel's tiles are 32 / 64 / 128 wide, so
the packed projection weights place every head in a 64-wide slot whose last 16 rows (q, k, v) / columns (out_proj) are
zero -- the scores and outputs are unchanged (zeros add exactly).

Token matrices are row-major [tokens * frames, C] with row = token * S + frame: attention "along time" is batch = token,
row stride 1; attention "along tracks" is batch = frame (batch stride one row), row stride S -- the kernels take strides,
nothing is permuted.

NB (modules.py:176-192): AttnBlock / CrossAttnBlock take their residual from the NORMALISED input, x = norm1(x); x = x +
attn(x): reproduced."""
import os

import torch
import torch.nn as nn

from ... import _C, graphs
from .. import convops as co

HEAD_SLOT = 64
# Linear layers over at least this many token rows run on the split-bf16 MFMA GEMM of the head convolutions (hi + lo
# operands, three MFMAs per product: 4e-6 of fp64, csrc/conv_igemm.hip) instead of the exact-fp32 MFMA kernel, which is
# built for a handful of rows (50 TF/s at 8 k rows against ~250).  0 = always exact fp32.
SPLIT_ROWS = int(os.environ.get("IGGT_TRACK_SPLIT_ROWS", "4096"))
_CONV_ACT = {None: 0, "gelu": 3}


def _p(t):
    return None if t is None else t.detach()


class _PackCache:
    """Derived weight layouts, rebuilt when the source parameters change (data_ptr / in-place version)."""

    def __init__(self):
        self._d = {}

    def get(self, key, params, make):
        sig = tuple((p.data_ptr(), p._version) for p in params)
        ent = self._d.get(key)
        if ent is None or ent[0] != sig:
            if ent is not None:
                graphs.buffers_changed()    # (the tracker runs eagerly, but its packs share the allocator with graphed buffers)
            ent = (sig, make())
            self._d[key] = ent
        return ent[1]


class PackedLinear:
    """y = act(x W^T + b) (+ res) for a fixed fp32 weight [N, K]: exact fp32 MFMA for few rows, split-bf16 MFMA for many."""

    def __init__(self, w, b):
        self.w = w.detach().float().contiguous()
        self.b = None if b is None else b.detach().float().contiguous()
        self._pc = None

    def __call__(self, x, act=None, res=None, out=None):
        M, K = x.shape
        N = self.w.shape[0]
        if (SPLIT_ROWS and M >= SPLIT_ROWS and K % 32 == 0 and N % 4 == 0 and act in _CONV_ACT and x.is_contiguous()
                and (res is None or res.is_contiguous()) and (out is None or out.is_contiguous())):
            if self._pc is None:
                self._pc = co.PackedConv(self.w, self.b, 1, 1, 1, 0, 0, K)
            y = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=x.device)
            co.run(self._pc, x.view(1, 1, M, K), out=y.view(1, 1, M, N), act=_CONV_ACT[act],
                   res=None if res is None else res.view(1, 1, M, N))
            return y
        return _C.linear_f32(x, self.w, self.b, act=act, res=res, out=out)


def pad_heads_rows(w, b, heads, d):
    """[heads * d, K] (+ bias) -> [heads * 64, K]: head h in rows [64 h, 64 h + d), zeros elsewhere."""
    K = w.shape[1]
    wp = torch.zeros(heads, HEAD_SLOT, K, dtype=torch.float32, device=w.device)
    wp[:, :d] = w.detach().float().view(heads, d, K)
    bp = torch.zeros(heads, HEAD_SLOT, dtype=torch.float32, device=w.device)
    bp[:, :d] = b.detach().float().view(heads, d)
    return wp.view(heads * HEAD_SLOT, K), bp.view(-1)


def pad_heads_cols(w, heads, d):
    """[N, heads * d] -> [N, heads * 64] with zero columns in the padding."""
    N = w.shape[0]
    wp = torch.zeros(N, heads, HEAD_SLOT, dtype=torch.float32, device=w.device)
    wp[:, :, :d] = w.detach().float().view(N, heads, d)
    return wp.view(N, heads * HEAD_SLOT).contiguous()


class Mlp(nn.Module):
    """fc1 -> exact GELU -> fc2 (modules.py:117-149; drop = 0, norm_layer unused by the tracker)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, norm_layer=None,
                 bias=True, drop=0.0, use_conv=False):
        super().__init__()
        if use_conv or act_layer is not nn.GELU:
            raise NotImplementedError("the tracker builds Linear + GELU MLPs only")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self._pk = _PackCache()

    def forward_rows(self, x, res=None, out=None):
        """x [M, K] fp32 (any row stride) -> fc2(gelu(fc1 x)) (+ res) [M, out]."""
        K, Kx = self.fc1.in_features, x.shape[1]

        def make():   # x rows may be zero-padded (aligned loads, K % 32 for the MFMA path): zero columns in the weight
            w1 = torch.nn.functional.pad(self.fc1.weight.detach().float(), (0, Kx - K))
            return PackedLinear(w1, self.fc1.bias), PackedLinear(self.fc2.weight, self.fc2.bias)

        l1, l2 = self._pk.get(Kx, (self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias), make)
        return l2(l1(x, act="gelu"), res=res, out=out)

    def forward(self, x):
        shp = x.shape
        return self.forward_rows(x.reshape(-1, shp[-1]).float().contiguous()).view(*shp[:-1], -1)


def _ln(norm: nn.LayerNorm, x, add=None):
    return _C.layernorm_rows(x, _p(norm.weight), _p(norm.bias), norm.eps, add=add)


class _MhaPacks:
    """Padded-head layouts of an nn.MultiheadAttention's projections."""

    def __init__(self, mha: nn.MultiheadAttention):
        self.mha = mha
        self._pk = _PackCache()

    def get(self):
        m = self.mha
        H, E = m.num_heads, m.embed_dim
        d = E // H

        def make():
            w, b = m.in_proj_weight, m.in_proj_bias
            parts = [pad_heads_rows(w[i * E:(i + 1) * E], b[i * E:(i + 1) * E], H, d) for i in range(3)]
            wq, bq = parts[0]
            wkv = torch.cat([parts[1][0], parts[2][0]], 0).contiguous()
            bkv = torch.cat([parts[1][1], parts[2][1]], 0).contiguous()
            wqkv = torch.cat([wq, wkv], 0).contiguous()
            bqkv = torch.cat([bq, bkv], 0).contiguous()
            wo = pad_heads_cols(m.out_proj.weight, H, d)
            return dict(q=PackedLinear(wq, bq), kv=PackedLinear(wkv, bkv), qkv=PackedLinear(wqkv, bqkv),
                        o=PackedLinear(wo, m.out_proj.bias), scale=d ** -0.5, HS=H * HEAD_SLOT)

        return self._pk.get(0, (m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias), make)


class AttnBlock(nn.Module):
    def __init__(self, hidden_size, num_heads, attn_class=nn.MultiheadAttention, mlp_ratio=4.0, **block_kwargs):
        super().__init__()
        self.norm1 = nn.LayerNorm(hidden_size)
        self.norm2 = nn.LayerNorm(hidden_size)
        self.attn = attn_class(embed_dim=hidden_size, num_heads=num_heads, batch_first=True, **block_kwargs)
        self.mlp = Mlp(in_features=hidden_size, hidden_features=int(hidden_size * mlp_ratio), drop=0)
        self._packs = _MhaPacks(self.attn)

    def forward_rows(self, x, batch, length, batch_stride, row_stride):
        """x [M, C] fp32 contiguous, updated in place.  Sequences: `batch` of them, `length` tokens each, token t of
        sequence b at row b * batch_stride + t * row_stride."""
        pk = self._packs.get()
        H, HS = self.attn.num_heads, pk["HS"]
        xn = _ln(self.norm1, x)
        qkv = pk["qkv"](xn)                                                    # [M, 3 HS]
        ao = torch.empty(x.shape[0], HS, dtype=torch.float32, device=x.device)
        ld = 3 * HS
        _C.attn_f32(qkv, qkv[:, HS:], qkv[:, 2 * HS:], ao, batch, H, length, length, HEAD_SLOT,
                    batch_stride * ld, row_stride * ld, batch_stride * ld, row_stride * ld, batch_stride * ld,
                    row_stride * ld, batch_stride * HS, row_stride * HS, pk["scale"])
        pk["o"](ao, res=xn, out=xn)                                           # x = norm1(x) + attn
        self.mlp.forward_rows(_ln(self.norm2, xn), res=xn, out=x)             # x = x + mlp(norm2 x)
        return x

    def forward(self, x, mask=None):
        """x [B, L, C] -> [B, L, C] (modules.py:176-192)."""
        B, L, C = x.shape
        y = x.reshape(B * L, C).float().contiguous().clone()
        return self.forward_rows(y, B, L, L, 1).view(B, L, C)


class CrossAttnBlock(nn.Module):
    def __init__(self, hidden_size, context_dim, num_heads=1, mlp_ratio=4.0, **block_kwargs):
        super().__init__()
        self.norm1 = nn.LayerNorm(hidden_size)
        self.norm_context = nn.LayerNorm(hidden_size)
        self.norm2 = nn.LayerNorm(hidden_size)
        self.cross_attn = nn.MultiheadAttention(embed_dim=hidden_size, num_heads=num_heads, batch_first=True,
                                                **block_kwargs)
        self.mlp = Mlp(in_features=hidden_size, hidden_features=int(hidden_size * mlp_ratio), drop=0)
        self._packs = _MhaPacks(self.cross_attn)

    def forward_rows(self, x, ctx, batch, len_q, len_k, q_strides, k_strides):
        """x [Mq, C] (updated in place), ctx [Mk, C]; *_strides = (batch stride, row stride) in rows."""
        pk = self._packs.get()
        H, HS = self.cross_attn.num_heads, pk["HS"]
        xn = _ln(self.norm1, x)
        cn = _ln(self.norm_context, ctx)
        q = pk["q"](xn)                                                       # [Mq, HS]
        kv = pk["kv"](cn)                                                     # [Mk, 2 HS]
        ao = torch.empty(x.shape[0], HS, dtype=torch.float32, device=x.device)
        _C.attn_f32(q, kv, kv[:, HS:], ao, batch, H, len_q, len_k, HEAD_SLOT,
                    q_strides[0] * HS, q_strides[1] * HS, k_strides[0] * 2 * HS, k_strides[1] * 2 * HS,
                    k_strides[0] * 2 * HS, k_strides[1] * 2 * HS, q_strides[0] * HS, q_strides[1] * HS, pk["scale"])
        pk["o"](ao, res=xn, out=xn)
        self.mlp.forward_rows(_ln(self.norm2, xn), res=xn, out=x)
        return x

    def forward(self, x, context, mask=None):
        """x [B, Lq, C], context [B, Lk, C] (modules.py:206-218); masks are never passed by the tracker."""
        if mask is not None:
            raise NotImplementedError("attention masks are unused by IGGT's tracker")
        B, Lq, C = x.shape
        Lk = context.shape[1]
        y = x.reshape(B * Lq, C).float().contiguous().clone()
        c = context.reshape(B * Lk, C).float().contiguous()
        return self.forward_rows(y, c, B, Lq, Lk, (Lq, 1), (Lk, 1)).view(B, Lq, C)
