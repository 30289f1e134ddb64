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
import torch
import torch.nn as nn

from ... import _C

HEAD_SLOT = 64


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
            ent = (sig, make())
            self._d[key] = ent
        return ent[1]


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
        K = self.fc1.in_features
        w1 = _p(self.fc1.weight)
        if x.shape[1] != K:     # rows padded to a multiple of 4 floats (aligned loads): zero columns in the weight
            w1 = self._pk.get("w1", (self.fc1.weight,), lambda: torch.nn.functional.pad(
                self.fc1.weight.detach().float(), (0, x.shape[1] - K)).contiguous())
        h = _C.linear_f32(x, w1, _p(self.fc1.bias), act="gelu")
        return _C.linear_f32(h, _p(self.fc2.weight), _p(self.fc2.bias), res=res, out=out)

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
            return dict(wq=wq.contiguous(), bq=bq.contiguous(), wkv=wkv, bkv=bkv, wqkv=wqkv, bqkv=bqkv, wo=wo,
                        bo=m.out_proj.bias.detach().float().contiguous(), scale=d ** -0.5, HS=H * HEAD_SLOT)

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
        qkv = _C.linear_f32(xn, pk["wqkv"], pk["bqkv"])                      # [M, 3 HS]
        ao = torch.empty(x.shape[0], HS, dtype=torch.float32, device=x.device)
        ld = 3 * HS
        _C.attn_f32(qkv, qkv[:, HS:], qkv[:, 2 * HS:], ao, batch, H, length, length, HEAD_SLOT,
                    batch_stride * ld, row_stride * ld, batch_stride * ld, row_stride * ld, batch_stride * ld,
                    row_stride * ld, batch_stride * HS, row_stride * HS, pk["scale"])
        _C.linear_f32(ao, pk["wo"], pk["bo"], res=xn, out=xn)                 # x = norm1(x) + attn
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
        q = _C.linear_f32(xn, pk["wq"], pk["bq"])                             # [Mq, HS]
        kv = _C.linear_f32(cn, pk["wkv"], pk["bkv"])                          # [Mk, 2 HS]
        ao = torch.empty(x.shape[0], HS, dtype=torch.float32, device=x.device)
        _C.attn_f32(q, kv, kv[:, HS:], ao, batch, H, len_q, len_k, HEAD_SLOT,
                    q_strides[0] * HS, q_strides[1] * HS, k_strides[0] * 2 * HS, k_strides[1] * 2 * HS,
                    k_strides[0] * 2 * HS, k_strides[1] * 2 * HS, q_strides[0] * HS, q_strides[1] * HS, pk["scale"])
        _C.linear_f32(ao, pk["wo"], pk["bo"], res=xn, out=xn)
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
