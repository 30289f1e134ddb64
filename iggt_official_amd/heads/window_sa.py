"""Window attention stages of the part head (reference iggt/heads/window_sa.py:26-545).

`SwinSA` = LayerNorm -> HAB (8x8 window self-attention, 4 heads x 32, no relative bias because the
reference passes the index tensor into an unused RoPE slot, appendix D.5; plus a 0.01-weighted
conv/channel-attention branch and an MLP) -> LayerNorm -> 3 convs.
`SwinCA` = the same wrapper around OCAB: 8x8 query windows attending to 12x12 overlapping key/value
windows (unfold stride 8, pad 2) with a learned relative-position bias.

Two reference behaviours are reproduced on purpose because trained weights depend on them:
  * OCAB query windows are cut from the *NCHW* tensor with an NHWC window routine
    (window_sa.py:280,286-287): `_ocab_query_windows` below applies the very same index
    permutation (appendix D.4);
  * map sizes must be multiples of the window (8): the reference fails in calculate_mask
    (window_sa.py:401-415); here a ValueError states the constraint.
Execution: NHWC fp32; every 3x3 convolution (CAB, conv_after_body, conv_before_upsample, conv_last) runs on
the implicit-GEMM MFMA kernel (csrc/conv_igemm.hip; CAB's 42-channel bottleneck is zero-padded to 64); LayerNorms
and Linear layers (q/k/v/proj, MLPs with fused GELU and residual adds) run on HIP kernels through heads/tokenops.py;
the window attention cores (8x8 queries; 64 keys, d = 32 in HAB; 144 keys, d = 64 and a relative-position bias in
OCAB) run on `iggt_window_attn_f32`, which gathers windows straight from the NHWC maps (no partition / Unfold copies).  Left on
PyTorch-ROCm ops: OCAB's query scramble (one permuted copy), CAB's squeeze-excite Linears, the residual adds.
"""
import torch
import torch.nn as nn

from .. import _C, profiling
from . import convops as co
from . import tokenops as tk
from .block import MemEffAttention


def window_partition(x, ws):
    b, h, w, c = x.shape
    x = x.view(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, c)


def window_reverse(windows, ws, h, w):
    b = windows.shape[0] // ((h // ws) * (w // ws))
    x = windows.view(b, h // ws, w // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, -1)


def _check_window_grid(h, w, ws):
    if h % ws or w % ws:
        raise ValueError(f"part-head window attention needs map sizes that are multiples of {ws}, got "
                         f"{h}x{w} (image H and W must be multiples of 28; reference window_sa.py:73,411)")


class ChannelAttention(nn.Module):
    def __init__(self, num_feat, squeeze_factor=16):
        super().__init__()
        self.attention = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(num_feat, num_feat // squeeze_factor, 1),
                                       nn.ReLU(inplace=True), nn.Conv2d(num_feat // squeeze_factor, num_feat, 1),
                                       nn.Sigmoid())

    def forward(self, x):
        return x * self.attention(x)


class CAB(nn.Module):
    def __init__(self, num_feat, compress_ratio=3, squeeze_factor=30):
        super().__init__()
        self.cab = nn.Sequential(nn.Conv2d(num_feat, num_feat // compress_ratio, 3, 1, 1), nn.GELU(),
                                 nn.Conv2d(num_feat // compress_ratio, num_feat, 3, 1, 1),
                                 ChannelAttention(num_feat, squeeze_factor))

        self._pk = co.PackCache()

    def forward(self, x):
        return self.cab(x)

    def forward_nhwc(self, x):
        """conv3x3 -> GELU -> conv3x3 -> channel attention (global average pool per frame), NHWC."""
        y, g = self.forward_nhwc_gate(x)
        return y * g[:, None, None, :]

    def forward_nhwc_gate(self, x):
        """The same as (map, per-frame channel gate): the caller folds the gate into its residual sum (one pass instead of three
        over the 8g map)."""
        c0, c2, ca = self.cab[0], self.cab[2], self.cab[3].attention
        mid = c0.out_channels
        mid_pad = (mid + 31) // 32 * 32
        p0 = self._pk.get(0, (c0.weight, c0.bias), lambda: co.pack_conv2d(c0))
        p2 = self._pk.get(2, (c2.weight, c2.bias), lambda: co.pack_conv2d(c2, cin_pad=mid_pad))
        t = co.run(p0, x, act=3, ldy=mid_pad, prec=co.PART_PREC)   # GELU fused; channels [mid, mid_pad) stay zero
        y = co.run(p2, t, prec=co.PART_PREC)
        g = y.mean(dim=(1, 2))                           # AdaptiveAvgPool2d(1) over the whole frame
        g = _C.linear_f32(g, ca[1].weight.detach().flatten(1), ca[1].bias.detach(), act="relu")      # squeeze  (fp32 HIP)
        g = _C.linear_f32(g, ca[3].weight.detach().flatten(1), ca[3].bias.detach(), act="sigmoid")   # excite
        return y, g


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x, res=None):
        """fc2(GELU(fc1(x))) (+ res): both Linears on the HIP kernel, GELU and the residual add fused (tokenops)."""
        act = 3 if isinstance(self.act, nn.GELU) else None
        if act is None:
            return tk.linear(self.fc2, self.act(tk.linear(self.fc1, x)), res=res)
        return tk.linear(self.fc2, tk.linear(self.fc1, x, act=act), res=res)


class _TokenNorm(nn.Module):
    """`patch_embed` of the reference wrapper: flatten NCHW to tokens + LayerNorm (window_sa.py:123-137)."""

    def __init__(self, embed_dim, norm_layer):
        super().__init__()
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        x = x.flatten(2).transpose(1, 2)
        return self.norm(x) if self.norm is not None else x


class HAB(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, compress_ratio=3,
                 squeeze_factor=30, conv_scale=0.01, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert shift_size == 0, "IGGT uses un-shifted windows only (window_sa.py:366)"
        self.dim, self.window_size, self.conv_scale = dim, window_size, conv_scale
        self.norm1 = norm_layer(dim)
        self.attn = MemEffAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.conv_block = CAB(num_feat=dim, compress_ratio=compress_ratio, squeeze_factor=squeeze_factor)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer)

    def forward(self, x, x_size, rpi_sa=None, attn_mask=None):
        h, w = x_size
        b, _, c = x.shape
        ws = self.window_size
        if ws != 8 or self.attn.head_dim not in (32, 64):
            raise _C.HipExtensionError("HIP window attention is built for 8x8 windows and head dim 32 / 64")
        y = tk.layer_norm(self.norm1, x).view(b, h, w, c)
        cab, gate = self.conv_block.forward_nhwc_gate(y.contiguous())
        # per-token Linear layers commute with the window partition: qkv / proj run on the whole map and the attention
        # kernel gathers the 8x8 windows itself (no window_partition / window_reverse copies)
        qkv = tk.linear(self.attn.qkv, y)                                                  # [b, h, w, 3c]
        o = torch.empty(b, h, w, c, dtype=torch.float32, device=x.device)
        # bench.py's part-branch leg: (kind, algorithmic FLOPs = 4 x windows x heads x 64 queries x keys x head dim,
        #                                algorithmic bytes = q, k, v read + o written, fp32)
        with profiling.region("window_attn", ("HAB 8x8 self", 4.0 * b * (h // 8) * (w // 8) * 64 * 64 * c, 16.0 * b * h * w * c)):
            _C.window_attn(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], o, self.attn.num_heads, self.attn.head_dim,
                           self.attn.scale)
        # x + att in the projection's epilogue, + conv_scale * gate * cab in ONE element-wise pass (window_sa.py:221-224 as three)
        xa = tk.linear(self.attn.proj, o.view(b, h * w, c), res=x)
        x = torch.addcmul(xa, cab.view(b, h * w, c), (gate * self.conv_scale)[:, None, :])
        if tk.mlp_h16_applicable(self.mlp.fc1, self.mlp.fc2, b * h * w):     # LayerNorm + MLP + residual on the 16-bit GEMM path, in place
            return tk.mlp_h16_(self.norm2, self.mlp.fc1, self.mlp.fc2, x)
        return self.mlp(tk.layer_norm(self.norm2, x), res=x)


def _ocab_query_windows(q_bhwc, ws):
    """Reference OCAB cuts query windows out of q.permute(0,3,1,2) (NCHW) with the NHWC partition
    routine and then reinterprets the buffer as [-1, ws*ws, c] (window_sa.py:280,286-287)."""
    b, h, w, c = q_bhwc.shape
    q = q_bhwc.permute(0, 3, 1, 2)                      # [b, c, h, w] read as (b, "h"=c, "w"=h, "c"=w)
    q = q.reshape(b, c // ws, ws, h // ws, ws, w).permute(0, 1, 3, 2, 4, 5)
    return q.reshape(-1, ws * ws, c)


class OCAB(nn.Module):
    def __init__(self, dim, input_resolution, window_size, overlap_ratio, num_heads, qkv_bias=True,
                 qk_scale=None, mlp_ratio=2, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.overlap_win_size = int(window_size * overlap_ratio) + window_size
        self.norm1 = norm_layer(dim)
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((window_size + self.overlap_win_size - 1) ** 2, num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.proj = nn.Linear(dim, dim)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=nn.GELU)

    def forward(self, x, k, v, x_size, rpi):
        h, w = x_size
        b, _, c = x.shape
        ws, ow, nh = self.window_size, self.overlap_win_size, self.num_heads
        d = c // nh
        shortcut = x
        if ws != 8 or d not in (32, 64):
            raise _C.HipExtensionError("HIP window attention is built for 8x8 windows and head dim 32 / 64")
        q = tk.linear(self.q, tk.layer_norm(self.norm1, x)).view(b, h, w, c)
        nk = tk.layer_norm(self.norm1, k)
        nv = nk if v is k else tk.layer_norm(self.norm1, v)        # the part head passes the same map as k and v (part_head.py:194)
        kk = tk.linear(self.k, nk).view(b, h, w, c)
        vv = tk.linear(self.v, nv).view(b, h, w, c)
        qw = _ocab_query_windows(q, ws).contiguous()                                 # [b*nw, 64, c] (layout quirk D.4)
        pad = (ow - ws) // 2
        # bias[head][key][query]: the kernel's lanes are the queries
        bias = self.relative_position_bias_table.detach().float()[rpi.reshape(-1)].view(ws * ws, ow * ow, nh)
        bias = bias.permute(2, 1, 0).contiguous()
        o = torch.empty(b, h, w, c, dtype=torch.float32, device=x.device)
        # keys / values: 12x12 windows at stride 8 read in place from the maps (zero vectors outside the image, like
        # nn.Unfold's padding); output written at the regular window positions (= window_reverse)
        # (bytes: q windows + o once, k and v once per 12 x 12 window that covers them = ow^2 / 64 times)
        with profiling.region("window_attn", ("OCAB 8x8 x 12x12 cross", 4.0 * b * (h // 8) * (w // 8) * 64 * ow * ow * c,
                                              4.0 * b * h * w * c * (2.0 + 2.0 * ow * ow / 64.0))):
            _C.window_attn(qw, kk, vv, o, nh, d, self.scale, q_windows=True, ow=ow, pad=pad, bias=bias)
        x = tk.linear(self.proj, o.view(b, h * w, c), res=shortcut)
        if tk.mlp_h16_applicable(self.mlp.fc1, self.mlp.fc2, b * h * w):
            return tk.mlp_h16_(self.norm2, self.mlp.fc1, self.mlp.fc2, x)
        return self.mlp(tk.layer_norm(self.norm2, x), res=x)


class SwinSA(nn.Module):
    def __init__(self, img_size=320, patch_size=1, out_chans=128, embed_dim=128, num_heads=6, window_size=16,
                 compress_ratio=3, squeeze_factor=30, conv_scale=0.01, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=nn.LayerNorm, patch_norm=True,
                 upscale=2, img_range=1.0, upsampler="pixelshuffle", resi_connection="1conv", _build_block=True):
        super().__init__()
        self.window_size = window_size
        self.register_buffer("relative_position_index_SA", self.calculate_rpi_sa())
        self.patch_embed = _TokenNorm(embed_dim, norm_layer if patch_norm else None)
        if _build_block:
            self.atten_block = HAB(dim=embed_dim, input_resolution=(img_size, img_size), num_heads=num_heads,
                                   window_size=window_size, shift_size=0, compress_ratio=compress_ratio,
                                   squeeze_factor=squeeze_factor, conv_scale=conv_scale, mlp_ratio=mlp_ratio,
                                   qkv_bias=qkv_bias, qk_scale=qk_scale, norm_layer=norm_layer)
        self.norm = norm_layer(embed_dim)
        self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1) if resi_connection == "1conv" else nn.Identity()
        self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, 64, 3, 1, 1), nn.LeakyReLU(inplace=True))
        self.conv_last = nn.Conv2d(64, out_chans, 3, 1, 1)

    def calculate_rpi_sa(self):
        ws = self.window_size
        coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        return rel.sum(-1)

    def _tail(self, body, x):
        """conv_after_body(body) + x -> conv3x3 -> LeakyReLU(0.01) -> conv3x3, all NHWC on the HIP conv kernel."""
        if not hasattr(self, "_pk"):
            self._pk = co.PackCache()
        cab, cbu, cl = self.conv_after_body, self.conv_before_upsample[0], self.conv_last
        p0 = self._pk.get(0, (cab.weight, cab.bias), lambda: co.pack_conv2d(cab))
        p1 = self._pk.get(1, (cbu.weight, cbu.bias), lambda: co.pack_conv2d(cbu))
        p2 = self._pk.get(2, (cl.weight, cl.bias), lambda: co.pack_conv2d(cl))
        y = co.run(p0, body, res=x, prec=co.PART_PREC)
        y = co.run(p1, y, act=2, prec=co.PART_PREC)
        return co.run(p2, y, prec=co.PART_PREC)

    def forward(self, x):
        """x NHWC -> NHWC."""
        x = x.contiguous()
        b, h, w, c = x.shape
        _check_window_grid(h, w, self.window_size)
        t = self.atten_block(tk.layer_norm(self.patch_embed.norm, x.view(b, h * w, c)), (h, w))
        body = tk.layer_norm(self.norm, t).view(b, h, w, c)
        return self._tail(body, x)


class SwinCA(SwinSA):
    def __init__(self, img_size=320, patch_size=1, out_chans=128, embed_dim=128, num_heads=6, window_size=16,
                 overlap_ratio=0.5, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                 norm_layer=nn.LayerNorm, patch_norm=True, upscale=2, img_range=1.0, upsampler="pixelshuffle",
                 resi_connection="1conv"):
        super().__init__(img_size=img_size, patch_size=patch_size, out_chans=out_chans, embed_dim=embed_dim,
                         num_heads=num_heads, window_size=window_size, norm_layer=norm_layer,
                         patch_norm=patch_norm, resi_connection=resi_connection, _build_block=False)
        self.overlap_ratio = overlap_ratio
        self.atten_block = OCAB(dim=embed_dim, input_resolution=(img_size, img_size), window_size=window_size,
                                overlap_ratio=overlap_ratio, num_heads=num_heads, qkv_bias=qkv_bias,
                                qk_scale=qk_scale, mlp_ratio=mlp_ratio, norm_layer=norm_layer)
        self.register_buffer("relative_position_index_OCA", self.calculate_rpi_oca())

    def calculate_rpi_oca(self):
        wo = self.window_size
        we = wo + int(self.overlap_ratio * wo)
        co = torch.stack(torch.meshgrid([torch.arange(wo), torch.arange(wo)], indexing="ij")).flatten(1)
        ce = torch.stack(torch.meshgrid([torch.arange(we), torch.arange(we)], indexing="ij")).flatten(1)
        rel = (ce[:, None, :] - co[:, :, None]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += wo - we + 1
        rel[:, :, 1] += wo - we + 1
        rel[:, :, 0] *= wo + we - 1
        return rel.sum(-1)

    def forward(self, x, k, v):
        """x, k, v NHWC -> NHWC."""
        x = x.contiguous()
        b, h, w, c = x.shape
        _check_window_grid(h, w, self.window_size)
        tn = lambda z: tk.layer_norm(self.patch_embed.norm, z.reshape(b, h * w, c))  # noqa: E731
        tkk = tn(k)
        tvv = tkk if v is k else tn(v)       # k and v are one tensor in the part head: normalise it once
        t = self.atten_block(tn(x), tkk, tvv, (h, w), self.relative_position_index_OCA)
        body = tk.layer_norm(self.norm, t).view(b, h, w, c)
        return self._tail(body, x)
