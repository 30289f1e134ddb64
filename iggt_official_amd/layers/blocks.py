"""Transformer block modules with the reference's names/parameters, executed by HIP kernels.

Mirrors reference iggt/layers/{attention.py:21-98, mlp.py:16-40, layer_scale.py:15-27,
block.py:27-107,210-259}.  Parameters stay fp32 `nn.Parameter`s under the reference's state-dict
keys (checkpoint contract); bf16 MFMA copies of the GEMM weights are packed
lazily (`packed()`), keyed on the parameter version so `load_state_dict` / `.to()` invalidate them.

Numerics: LayerNorm, q/k-norm, RoPE, LayerScale and the residual stream in fp32; GEMM and attention operands
16-bit with fp32 accumulate -- fp16 by default, bf16 (the reference's autocast GPU mode, demo.py:193-195) on request
(iggt_official_amd/precision.py).

Execution is in place on a flat fp32 token matrix x[T, C] (`Block.forward_inplace`); the
`forward(x, pos)` signatures of the reference are kept as thin wrappers.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import _C, graphs, precision, profiling


class LayerScale(nn.Module):
    def __init__(self, dim: int, init_values: float = 1e-5, inplace: bool = False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):  # only used outside the fused path
        return x * self.gamma


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0,
                 bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop = nn.Dropout(drop)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=True, proj_bias=True, attn_drop=0.0, proj_drop=0.0,
                 norm_layer=nn.LayerNorm, qk_norm=False, fused_attn=True, rope=None):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.fused_attn = fused_attn
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)
        self.proj_drop = nn.Dropout(proj_drop)
        self.rope = rope
        self.qk_norm = qk_norm


MemEffAttention = Attention  # reference attention.py:80-98 falls back to Attention.forward (xformers off)


# GEMM operands whose rows are a power of two apart (2 KiB for C = 1024) alias in the L2 when 256 rows x 64 B slices
# of many tiles stream in lockstep: the qkv GEMM (M=43968, N=3072, K=1024) measured 687 TF/s on dense operands and
# 823 TF/s with both row strides padded by 64 elements (probes/gemm_pad.py; proj / fc1 / fc2 are insensitive).
ROW_PAD = 64


class Workspace:
    """Re-usable device buffers for the block engine (sized for the largest T seen)."""

    def __init__(self):
        self._bufs = {}

    def get(self, name, shape, dtype, device):
        n = 1
        for s in shape:
            n *= s
        cur = self._bufs.get(name)
        if cur is None or cur.numel() < n or cur.dtype != dtype or cur.device != device:
            if cur is not None:
                graphs.buffers_changed()    # a captured graph may still point into the buffer that is dropped here
            cur = torch.empty(n, dtype=dtype, device=device)
            self._bufs[name] = cur
        return cur[:n].view(*shape)

    def get_padded(self, name, rows, cols, dtype, device, pad=ROW_PAD):
        """[rows, cols] view with a row stride of cols + pad elements (see ROW_PAD)."""
        return self.get(name, (rows, cols + pad), dtype, device)[:, :cols]


def _h16_weight(w: torch.Tensor, dt: torch.dtype, pad: int = 0):
    w = w.detach().to(dt)
    if pad == 0:
        return w.contiguous()
    buf = torch.zeros(w.shape[0], w.shape[1] + pad, dtype=dt, device=w.device)
    buf[:, :w.shape[1]] = w
    return buf[:, :w.shape[1]]


def _h16_residual(w: torch.Tensor, dt: torch.dtype):
    """dW = W - round16(W), itself stored in 16 bits (mean-input compensation, csrc/elementwise.hip)."""
    w = w.detach().float()
    return (w - w.to(dt).float()).to(dt).contiguous()


GEMM_MAX_OPERAND_ELEMENTS = 1 << 31   # csrc/gemm_bf16_t256.hip, gemm_bf16_duo.hip: M x lda and N x ldw below this


def _gemm_rows(a, w, out, **kw):
    """_C.gemm_h16 over row chunks when M x lda would leave the 32-bit operand offsets of the LDS-DMA kernels (the K = 3 x 4 096
    GEMM of the x3 rung at 64 views @ 1036^2: 350 784 rows x 12 288 columns); rows are independent, the chunks run back to back."""
    M, lda = a.shape[0], a.stride(0)
    if M * lda < GEMM_MAX_OPERAND_ELEMENTS:
        return _C.gemm_h16(a, w, out, **kw)
    step = max(256, (GEMM_MAX_OPERAND_ELEMENTS - 1) // lda // 256 * 256)
    for r0 in range(0, M, step):
        _C.gemm_h16(a[r0:r0 + step], w, out[r0:r0 + step], **kw)
    return out


def _x3_weight(w: torch.Tensor):
    """W' = [W_hi | W_hi | W_lo] (fp16, [N, 3 K]) for the three-pass GEMM of the x3 precision rung against A' = [A_hi | A_lo | A_hi]
    (csrc/x3.hip): A W^T ~= A_hi W_hi^T + A_lo W_hi^T + A_hi W_lo^T."""
    w = w.detach().float()
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    return torch.cat([hi, hi, lo], 1).contiguous()


# Range folding (round 4).  fp16 operands hold |w| <= 65504; a checkpoint whose weights exceed that used to be sent to bf16
# operands as a whole -- the mode that sits 6e-3 from the fp32 reference.  A block's four GEMMs sit between per-channel affines
# that commute with a power-of-two rescaling EXACTLY (no rounding: only exponents change):
#   qkv / fc1 input column c   <->  norm1 / norm2 weight and bias of channel c   (x_n W^T = (x_n s)(W / s)^T)
#   proj / fc2 output row n    <->  LayerScale gamma_n and the bias b_n          (gamma (W x + b) = (gamma s)((W / s) x + b / s))
#   proj input column n        <->  row n of the V third of qkv and its bias     (attention is linear in V)
# so an outlying column / row is brought back to O(1) at pack time and its partner absorbs the factor.  This is the
# re-parametrisation freedom a trained checkpoint may sit anywhere in (a LayerNorm scale of 1e-5 in front of weights of 1e5);
# the activations between the two partners then have ordinary magnitudes as well.  Only slices whose largest entry exceeds
# FOLD_HI are touched, so ordinary checkpoints pack bit-identically to round 3.  What cannot be folded (rows of qkv's Q / K
# thirds and of fc1, columns of fc2: a non-linearity follows) and still exceeds the range sends THAT BLOCK to bf16 operands.
FOLD_HI = 1024.0
PARTNER_MAX = 1024.0   # largest LayerNorm scale / bias a fold may leave behind (the normalised activations are O(10) x that, in fp16)


def _pow2_scale(w: torch.Tensor, dim: int):
    """Per-slice power of two for the slices of `w` along `dim` (dim = 0: columns, 1: rows) that are outliers AS A WHOLE: the
    decision uses the slice's MEDIAN magnitude, not its maximum -- an outlying row raises the maximum of every column it
    crosses (and vice versa) but not their medians, and folding the wrong partner would push ordinary entries into fp16's
    subnormal range.  A slice is rescaled when its maximum exceeds FOLD_HI and its median is > 64x the typical slice median; the
    factor brings its median back to the typical one."""
    a = w.abs()
    med = a.median(dim=dim).values.clamp_min(1e-30)
    typical = med.median().clamp_min(1e-30)
    s = torch.exp2(torch.round(torch.log2(med / typical)))
    hit = (a.amax(dim=dim) > FOLD_HI) & (med > 64.0 * typical)
    return torch.where(hit, s, torch.ones_like(s))


def fold_ranges(wq, bq, wp, bp, w1, w2, b2, n1w, n1b, n2w, n2b, g1, g2):
    """Exact power-of-two re-parametrisation of one block (fp32 tensors in, fp32 tensors out; see FOLD_HI above).  Returns the
    thirteen tensors in the same order plus the number of slices that were rescaled."""
    C = wp.shape[0]
    touched = 0
    # qkv input columns <-> norm1
    s = _pow2_scale(wq, 0)
    touched += int((s != 1).sum())
    wq, n1w, n1b = wq / s[None, :], n1w * s, n1b * s
    # proj input columns <-> V rows of qkv
    r = _pow2_scale(wp, 0)
    touched += int((r != 1).sum())
    wp = wp / r[None, :]
    wq = torch.cat([wq[:2 * C], wq[2 * C:] * r[:, None]], 0)
    if bq is not None:
        bq = torch.cat([bq[:2 * C], bq[2 * C:] * r], 0)
    # proj output rows <-> LayerScale 1
    s = _pow2_scale(wp, 1)
    touched += int((s != 1).sum())
    wp, g1 = wp / s[:, None], g1 * s
    if bp is not None:
        bp = bp / s
    # fc1 input columns <-> norm2
    s = _pow2_scale(w1, 0)
    touched += int((s != 1).sum())
    w1, n2w, n2b = w1 / s[None, :], n2w * s, n2b * s
    # fc2 output rows <-> LayerScale 2
    s = _pow2_scale(w2, 1)
    touched += int((s != 1).sum())
    w2, g2 = w2 / s[:, None], g2 * s
    if b2 is not None:
        b2 = b2 / s
    return wq, bq, wp, bp, w1, w2, b2, n1w, n1b, n2w, n2b, g1, g2, touched


MEAN_SAMPLE_ROWS = 1024  # the column mean is taken over ~this many evenly spaced rows (sampling error sigma / 32)


def compensated_bias(ws: "Workspace", a: torch.Tensor, dw: Optional[torch.Tensor], bias: Optional[torch.Tensor]):
    """bias + dW mean_rows(a): restores the part of the weight rounding that is common to all tokens
    (precision.py); `bias` itself when compensation is off.  Two small launches (column mean over ~MEAN_SAMPLE_ROWS evenly spaced
    rows, matrix-vector product); one fused launch with an in-kernel hand-over was built in round 6 and measured slower
    (csrc/elementwise.hip, profiles/r06_comp_bias_ab.txt)."""
    if dw is None:
        return bias
    N, K = dw.shape
    mu = ws.get("mean_in", (K,), torch.float32, a.device)
    out = ws.get("bias_comp", (N,), torch.float32, a.device)
    _C.colmean(a, mu, max(1, a.shape[0] // MEAN_SAMPLE_ROWS))
    return _C.bias_correct(dw, mu, bias, out)


class Block(nn.Module):
    """x += ls1(attn(norm1(x))); x += ls2(mlp(norm2(x)))   (reference block.py:105-106)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, proj_bias=True, ffn_bias=True, drop=0.0,
                 attn_drop=0.0, init_values=None, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 attn_class=Attention, ffn_layer=Mlp, qk_norm=False, fused_attn=True, rope=None):
        super().__init__()
        self.dim = dim
        self.norm1 = norm_layer(dim)
        self.attn = attn_class(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias,
                               attn_drop=attn_drop, proj_drop=drop, qk_norm=qk_norm, fused_attn=fused_attn,
                               rope=rope)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = ffn_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer,
                             drop=drop, bias=ffn_bias)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.sample_drop_ratio = drop_path
        self._packed = None
        self._packed_key = None
        # host-side switch of the estimated-shift launches of this block's attention (csrc/attention_est.hip): off until a
        # snapshot of the guard word shows that the norm bound flagged tiles here (models/aggregator.py _apply_guard_snapshot).
        # While it is off the block issues exactly the round-3 launch sequence: the nine extra launches of the estimated-shift
        # sequence cost ~45 us per attention call even when every one of them returns at once (2 ms per 32-view forward).
        self._est_on = False
        # precision rung (precision.py "x3"): the owner of a block SEQUENCE (models/aggregator.py plan_escalation) asks for the x3
        # path here when this block or ANY LATER block of the sequence is ill-conditioned by its own figures -- an ill-conditioned
        # block amplifies the operand rounding of everything upstream of it, so escalating it alone does not help (CPU
        # simulation, profiles/r05_escalation_policy.txt: sigma_qk = 1 with only the sharp-softmax blocks escalated 5.2e-3, with
        # all of them 1.8e-5).  None: a block used on its own decides from its own figures.
        self._x3_request: Optional[bool] = None
        self._cond = None
        self._cond_key = None

    # ------------------------------------------------------------------------------------------
    def own_condition(self) -> dict:
        """precision.block_condition of this block's parameters as stored (cached per parameter version; the first call after a
        change reads a few scalars back from the device)."""
        ps = [self.norm1.weight, self.norm2.weight]
        if self.attn.qk_norm:
            ps += [self.attn.q_norm.weight, self.attn.k_norm.weight]
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)
        if self._cond_key != key:
            self._cond = precision.block_condition(ps[0], ps[1], ps[2] if self.attn.qk_norm else None,
                                                   ps[3] if self.attn.qk_norm else None, self.attn.scale)
            self._cond_key = key
        return self._cond

    def own_escalation(self) -> bool:
        return precision.should_escalate(self.own_condition())

    def packed(self):
        """16-bit (precision.operand_dtype()) copies of the four GEMM weights + fp32 epilogue vectors, rebuilt when
        params or the operand format change."""
        ps = (self.attn.qkv.weight, self.attn.proj.weight, self.mlp.fc1.weight, self.mlp.fc2.weight)
        dt = precision.operand_dtype()
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in ps) + (dt, precision.mean_compensation_sites(),
                                                                              precision.range_folding(), precision.escalation(),
                                                                              self._x3_request)
        if self._packed_key != key:
            if self._packed is not None:
                graphs.buffers_changed()    # the old packs are freed below; captured graphs hold their addresses
            dev = ps[0].device
            f32 = lambda t: None if t is None else t.detach().float().contiguous()  # noqa: E731
            ones = lambda: torch.ones(self.dim, device=dev)  # noqa: E731
            wq, wp, w1, w2 = (f32(p) for p in ps)
            bq, bp, b1, b2 = (f32(l.bias) for l in (self.attn.qkv, self.attn.proj, self.mlp.fc1, self.mlp.fc2))
            g1 = f32(self.ls1.gamma) if isinstance(self.ls1, LayerScale) else ones()
            g2 = f32(self.ls2.gamma) if isinstance(self.ls2, LayerScale) else ones()
            n1w, n1b, n2w, n2b = f32(self.norm1.weight), f32(self.norm1.bias), f32(self.norm2.weight), f32(self.norm2.bias)
            folded = 0
            if dt == torch.float16:
                raw = float(torch.stack([t.abs().amax() for t in (wq, wp, w1, w2)]).amax())
                if raw != raw:       # NaN: nothing downstream can be right; say so instead of quietly picking another format
                    raise ValueError("a transformer block holds NaN weights")
            if dt == torch.float16 and precision.range_folding():
                # one device-side reduction decides whether anything has to be folded at all (the common case: nothing)
                if raw > FOLD_HI:
                    unfolded = (wq, bq, wp, bp, w1, w2, b2, n1w, n1b, n2w, n2b, g1, g2)
                    wq, bq, wp, bp, w1, w2, b2, n1w, n1b, n2w, n2b, g1, g2, folded = fold_ranges(*unfolded)
                    # the PARTNERS absorb the factors and feed fp16 activations themselves (LayerNorm output = |.| <~ 10 x scale;
                    # v = LayerNorm output x V rows): a slice that is genuinely large -- its partner NOT correspondingly small --
                    # would turn them into 1e6 and the stores into +-65504 / the GEMMs into inf.  Then nothing is
                    # folded: the weights run as stored if they fit fp16, else this block falls to bf16 below.
                    part = float(torch.stack([t.abs().amax() for t in (n1w, n1b, n2w, n2b)]).amax())
                    part0 = float(torch.stack([t.abs().amax() for t in (unfolded[7], unfolded[8], unfolded[9], unfolded[10])]).amax())
                    vmax, vmax0 = float(wq[2 * self.dim:].abs().amax()), float(unfolded[0][2 * self.dim:].abs().amax())
                    if (part > PARTNER_MAX and part > part0) or (vmax > FOLD_HI and vmax > vmax0):
                        wq, bq, wp, bp, w1, w2, b2, n1w, n1b, n2w, n2b, g1, g2 = unfolded
                        folded = 0
            if dt == torch.float16:
                worst = float(torch.stack([t.abs().amax() for t in (wq, wp, w1, w2)]).amax())
                if not (worst <= 65504.0):
                    if not precision.range_folding():
                        precision.check_operand_range("block weight", torch.tensor(worst), dt)   # raises (round-3 behaviour)
                    # what is left beyond the range has no exact partner to fold into: this block alone runs on bf16 operands
                    import logging

                    logging.getLogger(__name__).warning(
                        "a transformer block keeps max |w| = %.4g after range folding: it runs on bf16 operands (the others "
                        "stay on fp16)", worst)
                    dt = torch.bfloat16
                    wq, wp, w1, w2 = (f32(p) for p in ps)
                    bq, bp, b2 = f32(self.attn.qkv.bias), f32(self.attn.proj.bias), f32(self.mlp.fc2.bias)
                    g1 = f32(self.ls1.gamma) if isinstance(self.ls1, LayerScale) else ones()
                    g2 = f32(self.ls2.gamma) if isinstance(self.ls2, LayerScale) else ones()
                    n1w, n1b = f32(self.norm1.weight), f32(self.norm1.bias)
                    n2w, n2b = f32(self.norm2.weight), f32(self.norm2.bias)
                    folded = 0
            cont = lambda t: None if t is None else t.contiguous()  # noqa: E731
            # per-block precision rung (precision.py "x3"): decided from this block's own LayerNorm / q-k-norm scales as stored
            cond = self.own_condition()
            x3 = self.own_escalation() if self._x3_request is None else (self._x3_request and precision.escalation() != "off")
            if dt == torch.float16 and x3:
                self._packed = dict(
                    x3=True, condition=cond, folded_slices=folded,
                    w3_qkv=_x3_weight(wq), b_qkv=cont(bq), w3_proj=_x3_weight(wp), b_proj=cont(bp),
                    w3_fc1=_x3_weight(w1), b_fc1=cont(b1), w3_fc2=_x3_weight(w2), b_fc2=cont(b2),
                    g1=g1.contiguous(), g2=g2.contiguous(),
                    n1w=n1w.contiguous(), n1b=n1b.contiguous(), n2w=n2w.contiguous(), n2b=n2b.contiguous())
                if self.attn.qk_norm:
                    self._packed.update(qw=f32(self.attn.q_norm.weight), qb=f32(self.attn.q_norm.bias),
                                        kw=f32(self.attn.k_norm.weight), kb=f32(self.attn.k_norm.bias))
                self._est_on = False
                self._packed_key = key
                return self._packed
            self._packed = dict(
                w_qkv=_h16_weight(wq, dt, ROW_PAD), b_qkv=cont(bq),
                w_proj=_h16_weight(wp, dt), b_proj=cont(bp),
                w_fc1=_h16_weight(w1, dt), b_fc1=cont(b1),
                w_fc2=_h16_weight(w2, dt), b_fc2=cont(b2),
                g1=g1.contiguous(), g2=g2.contiguous(),
                n1w=n1w.contiguous(), n1b=n1b.contiguous(), n2w=n2w.contiguous(), n2b=n2b.contiguous(),
                folded_slices=folded, x3=False, condition=cond, bf16_fallback=(dt != precision.operand_dtype()),
            )
            comp = precision.mean_compensation_sites() if dt == torch.float16 else frozenset()
            for n, w_ in (("qkv", wq), ("proj", wp), ("fc1", w1), ("fc2", w2)):
                self._packed["dw_" + n] = _h16_residual(w_, dt) if n in comp else None
            if self.attn.qk_norm:
                self._packed.update(qw=f32(self.attn.q_norm.weight), qb=f32(self.attn.q_norm.bias),
                                    kw=f32(self.attn.k_norm.weight), kb=f32(self.attn.k_norm.bias))
                # adaptive switch of the static-bound attention (include/iggt_hip.h): persistent per call site, reset with the
                # packs (new weights -> new score statistics)
                self._packed["guard"] = _C.new_attn_guard(dev)
                self._est_on = False
            self._packed_key = key
        return self._packed

    # ------------------------------------------------------------------------------------------
    def forward_inplace(self, x2d: torch.Tensor, ws: Workspace, *, batch: int, tokens: int,
                        rope_geom: Optional[dict] = None, kv_gather=None, q_rows_per_wg: int = 0,
                        guard_prev: Optional[torch.Tensor] = None):
        """Run the block in place on x2d [T, C] fp32 (T = batch * tokens rows; attention is computed
        independently per `batch` group of `tokens` rows).

        rope_geom: dict(P=tokens per view, gw=grid width, patch_start=5, cos=..., sin=...) when the block
        has q/k-norm + RoPE (aggregator blocks); None for the DINOv2 blocks.
        kv_gather: multi-GPU global attention: a dist.ViewShard (K/V all-gather pipelined over head groups when its
        kv_groups > 1) or a callable(kv_local [T, 2C]) -> kv_all [T_all, 2C]; None on a single GPU.
        guard_prev: adaptive-switch word of the same kind of block one layer earlier (`attn_guard()`), consulted only while
        this block has never been measured.
        """
        if x2d.dtype != torch.float32 or x2d.dim() != 2 or x2d.stride(1) != 1:
            raise _C.HipExtensionError("Block.forward_inplace expects a row-major fp32 [T, C] matrix")
        if self.attn.head_dim != 64:
            raise _C.HipExtensionError("HIP flash attention is built for head_dim 64")
        T, C = x2d.shape
        assert T == batch * tokens and C == self.dim
        dev = x2d.device
        pk = self.packed()
        H = self.attn.num_heads
        if pk["x3"]:
            return self._forward_x3(x2d, ws, pk, batch=batch, tokens=tokens, rope_geom=rope_geom, kv_gather=kv_gather)
        dt = pk["w_qkv"].dtype  # 16-bit operand format of this block (bf16 for a block whose weights could not be folded)
        alt = "" if dt == precision.operand_dtype() else "_alt"   # its own buffers: no reallocation when neighbours differ
        xn = ws.get_padded("xn" + alt, T, C, dt, dev)
        qkv = ws.get("qkv" + alt, (T, 3 * C), dt, dev)
        ao = ws.get("ao" + alt, (T, C), dt, dev)
        hid = ws.get("hid" + alt, (T, pk["w_fc1"].shape[0]), dt, dev)

        sat = precision.debug_saturation()
        _C.layernorm(x2d, pk["n1w"], pk["n1b"], xn, self.norm1.eps)
        b_ = compensated_bias(ws, xn, pk["dw_qkv"], pk["b_qkv"])
        with profiling.region("gemm", ("qkv", T, 3 * C, C)):       # bench.py's secondary roofline leg: (name, M, N, K)
            _C.gemm_h16(xn, pk["w_qkv"], qkv, bias=b_)
        if sat:
            precision.count_saturation("norm1", xn)
            precision.count_saturation("qkv", qkv)
        k_src, v_src, kv_rs, Nk, k_bs = qkv[:, C:], qkv[:, 2 * C:], 3 * C, tokens, tokens * 3 * C
        grouped = None
        overlapped = False
        # static-bound softmax (csrc/attention_v3.hip): q leaves the q/k-norm kernel pre-scaled by scale * log2 e together
        # with the per-head maxima of |q| and |k|; not for the head-group pipelined gather (per-group launches)
        static = (self.attn.qk_norm and precision.static_softmax() and H == 16
                  and getattr(kv_gather, "kv_groups", 1) <= 1)
        qkmax = ws.get("qkmax", (_C.QKMAX_NUMEL,), torch.float32, dev) if static else None
        sk = dict(q_scale=self.attn.scale * _C.LOG2E, qkmax=qkmax) if static else {}
        guard = pk["guard"] if (static and precision.static_guard()) else None
        if guard is None:
            guard_prev = None
        # estimated-shift launches (csrc/attention_est.hip) are a one-pass form: a call site that needs them gives up the key-range
        # split of small grids and, in a view-sharded run, the overlap of the K/V gather with the own-key attention (round 5:
        # before, a sharded run on adversarial score statistics fell to the online-max kernel on every rank)
        use_est = guard is not None and self._est_on and precision.attn_estimated_shift() and q_rows_per_wg == 0
        if self.attn.qk_norm:
            assert rope_geom is not None
            qk_args = (pk["qw"], pk["qb"], pk["kw"], pk["kb"], rope_geom["cos"], rope_geom["sin"], T, rope_geom["P"],
                       rope_geom["gw"], rope_geom["patch_start"], self.attn.q_norm.eps)
            if kv_gather is None:
                _C.qknorm_rope(qkv, qkv, qkv[:, C:], None, *qk_args, **sk)
            elif getattr(kv_gather, "kv_groups", 1) > 1:
                # multi-GPU, pipelined over head groups (dist.py): K|V of head group g in kv_local[g]
                assert batch == 1
                G = kv_gather.kv_groups
                hg = H // G
                D = 2 * hg * 64
                kv_local = ws.get("kv_local" + alt, (G, T, D), dt, dev)
                _C.qknorm_rope(qkv, qkv, kv_local[0], kv_local[0][:, hg * 64:], *qk_args, heads_per_group=hg,
                               k_group_stride=T * D, v_group_stride=T * D)
                grouped = (G, hg, D, kv_gather.gather_kv_groups(kv_local))
            else:
                gather = kv_gather.all_gather_kv if hasattr(kv_gather, "all_gather_kv") else kv_gather
                kv_local = ws.get("kv_local" + alt, (T, 2 * C), dt, dev)
                _C.qknorm_rope(qkv, qkv, kv_local, kv_local[:, C:], *qk_args, **sk)
                assert batch == 1
                if (static and q_rows_per_wg == 0 and hasattr(kv_gather, "all_gather_kv_begin") and kv_gather.active
                        and not use_est):
                    overlapped = self._attend_overlapped(qkv, kv_local, kv_gather, qkmax, ao, ws, T, H, C, guard, guard_prev,
                                                         overlap=precision.gather_overlap())
                elif (static and q_rows_per_wg == 0 and hasattr(kv_gather, "all_gather_kv_begin") and kv_gather.active):
                    # a sharded call site on the estimated shift: gather first, then the one-pass launch below.  The SAME two
                    # collectives as the overlapped form (K/V rows, then the 32 norm maxima) -- `use_est` is a rank-LOCAL decision
                    # (each rank's own guard words), and ranks that disagree on it must still issue identical collectives; the
                    # gathered maxima also give the key bound over all ranks without a pass over the gathered rows
                    kv_all, stats_all = kv_gather.all_gather_kv(kv_local, qkmax[:32])
                    qkmax[16:32].copy_(stats_all[:, 16:32].amax(0))
                    k_src, v_src, kv_rs, Nk, k_bs = kv_all, kv_all[:, C:], 2 * C, kv_all.shape[0], 0
                else:
                    kv_all = gather(kv_local)
                    k_src, v_src, kv_rs, Nk, k_bs = kv_all, kv_all[:, C:], 2 * C, kv_all.shape[0], 0
                    if static:   # the key bound has to cover every rank's keys: measured on the gathered rows
                        _C.k_rownorm_max(kv_all[:, :C], qkmax)
        elif kv_gather is not None:
            raise _C.HipExtensionError("kv_gather needs a q/k-norm block")
        if overlapped:
            pass
        elif grouped is None:
            with profiling.region("global_attn" if batch == 1 else "frame_attn", (batch, tokens, Nk)):
                if static:
                    flags = ws.get("attn_flags", (batch * H * ((tokens + 127) // 128),), torch.int32, dev)
                    nws = _C.static_attn_ws_bytes(batch, H, tokens, Nk) if (q_rows_per_wg == 0 and not use_est) else 0
                    part_ws = ws.get("attn_part", (nws,), torch.uint8, dev) if nws else None
                    est_ws = None
                    if use_est:
                        est_ws = ws.get("attn_est", (_C.static_attn_est_ws_bytes(batch, H, tokens, Nk),), torch.uint8, dev)
                    # the estimated-shift pre-pass samples the special tokens of every view among the keys: the first
                    # `patch_start` rows of every P rows (frame attention: of the one view; global: of each view)
                    _C.flash_attn_d64_static(qkv, k_src, v_src, ao, batch, H, tokens, Nk,
                                             tokens * 3 * C, 3 * C, k_bs, kv_rs, k_bs, kv_rs, tokens * C, C,
                                             qkmax, flags, q_rows_per_wg, part_ws, guard, guard_prev, est_ws=est_ws,
                                             key_period=rope_geom["P"], key_nspecial=rope_geom["patch_start"])
                else:
                    _C.flash_attn_d64(qkv, k_src, v_src, ao, batch, H, tokens, Nk,
                                      tokens * 3 * C, 3 * C, k_bs, kv_rs, k_bs, kv_rs, tokens * C, C,
                                      self.attn.scale, q_rows_per_wg)
        else:
            G, hg, D, handles = grouped
            main = torch.cuda.current_stream()
            Nk = handles[0][1].shape[0]
            with profiling.region("global_attn", (1, tokens, Nk)):   # includes the overlapped wait for the gather
                ready = kv_gather.event(G)
                ready.record(main)                                    # q (and kv_local) are final on `main`
                for g, (work, kv_all) in enumerate(handles):
                    st = kv_gather.side_stream(g)
                    st.wait_event(ready)
                    with torch.cuda.stream(st):
                        if work is not None:
                            work.wait()                               # this stream waits for collective g only
                        _C.flash_attn_d64(qkv[:, g * hg * 64:], kv_all, kv_all[:, hg * 64:], ao[:, g * hg * 64:],
                                          1, hg, tokens, Nk, 0, 3 * C, 0, D, 0, D, 0, C, self.attn.scale,
                                          q_rows_per_wg)
                        kv_gather.event(g).record(st)
                for g in range(G):
                    main.wait_event(kv_gather.event(g))
        if sat:
            precision.count_saturation("attn_out", ao)
        b_ = compensated_bias(ws, ao, pk["dw_proj"], pk["b_proj"])
        with profiling.region("gemm", ("proj", T, C, C)):
            _C.gemm_h16(ao, pk["w_proj"], x2d, bias=b_, gamma=pk["g1"], accumulate=True)
        _C.layernorm(x2d, pk["n2w"], pk["n2b"], xn, self.norm2.eps)
        b_ = compensated_bias(ws, xn, pk["dw_fc1"], pk["b_fc1"])
        with profiling.region("gemm", ("fc1", T, hid.shape[1], C)):
            _C.gemm_h16(xn, pk["w_fc1"], hid, bias=b_, act=1)
        if sat:
            precision.count_saturation("norm2", xn)
            precision.count_saturation("mlp_hidden", hid)
        b_ = compensated_bias(ws, hid, pk["dw_fc2"], pk["b_fc2"])
        with profiling.region("gemm", ("fc2", T, C, hid.shape[1])):
            _C.gemm_h16(hid, pk["w_fc2"], x2d, bias=b_, gamma=pk["g2"], accumulate=True)
        return x2d

    def _forward_x3(self, x2d, ws, pk, *, batch, tokens, rope_geom, kv_gather):
        """The block on fp16 hi + lo operand PAIRS, three MFMA passes per product (precision.py "x3", csrc/x3.hip): same data
        flow as forward_inplace, every 16-bit rounding site replaced by a 22-bit one.
          LayerNorm -> [hi | lo | hi]  --GEMM K = 3C, fp32 out-->  qkv fp32  -> q/k-norm + RoPE + split -> attention on pairs ->
          [hi | lo | hi]  --GEMM, LayerScale + residual-->  x;   LayerNorm -> [hi | lo | hi] -> fc1 fp32 -> exact GELU + split ->
          fc2, LayerScale + residual.
        No mean-input compensation (the weights are pairs too), no static softmax (online maximum in fp32).  Multi-GPU: K and V
        pairs travel as one [T_local, 4C] message, the gather completes before the one attention launch."""
        T, C = x2d.shape
        dev = x2d.device
        H = self.attn.num_heads
        Hd = pk["w3_fc1"].shape[0]
        f16 = torch.float16
        xn3 = ws.get("x3_xn", (T, 3 * C), f16, dev)
        qkv32 = ws.get("x3_qkv32", (T, 3 * C), torch.float32, dev)
        _C.layernorm(x2d, pk["n1w"], pk["n1b"], xn3, self.norm1.eps, split3=True)
        with profiling.region("gemm", ("qkv_x3", T, 3 * C, 3 * C)):
            _gemm_rows(xn3, pk["w3_qkv"], qkv32, bias=pk["b_qkv"])
        q_scale = self.attn.scale * _C.LOG2E
        norm = dict(qw=pk["qw"], qb=pk["qb"], kw=pk["kw"], kb=pk["kb"], eps=self.attn.q_norm.eps) if self.attn.qk_norm else {}
        rope = {}
        if self.attn.qk_norm:
            assert rope_geom is not None
            rope = dict(cos_t=rope_geom["cos"], sin_t=rope_geom["sin"], P=rope_geom["P"], gw=rope_geom["gw"],
                        patch_start=rope_geom["patch_start"])
        if kv_gather is None:
            qkv6 = ws.get("x3_qkv6", (T, 6 * C), f16, dev)     # [q_hi | k_hi | v_hi | q_lo | k_lo | v_lo]
            _C.qkv_split(qkv32, qkv6[:, :C], 3 * C, qkv6[:, C:2 * C], 3 * C, qkv6[:, 2 * C:3 * C], 3 * C, q_scale=q_scale,
                         **norm, **rope)
            q, q_lo, q_rs, q_bs = qkv6, qkv6[:, 3 * C:], 6 * C, tokens * 6 * C
            k, k_lo, v, v_lo = qkv6[:, C:], qkv6[:, 4 * C:], qkv6[:, 2 * C:], qkv6[:, 5 * C:]
            kv_rs, Nk, k_bs = 6 * C, tokens, tokens * 6 * C
        else:
            if not self.attn.qk_norm or batch != 1:
                raise _C.HipExtensionError("kv_gather needs a q/k-norm block")
            q2 = ws.get("x3_q2", (T, 2 * C), f16, dev)                 # [q_hi | q_lo]
            kv_local = ws.get("x3_kv_local", (T, 4 * C), f16, dev)     # [k_hi | v_hi | k_lo | v_lo]: one gather message
            _C.qkv_split(qkv32, q2[:, :C], C, kv_local[:, :C], 2 * C, kv_local[:, C:2 * C], 2 * C, q_scale=q_scale, **norm,
                         **rope)
            gather = kv_gather.all_gather_kv if hasattr(kv_gather, "all_gather_kv") else kv_gather
            kv_all = gather(kv_local)
            q, q_lo, q_rs, q_bs = q2, q2[:, C:], 2 * C, 0
            k, k_lo, v, v_lo = kv_all, kv_all[:, 2 * C:], kv_all[:, C:], kv_all[:, 3 * C:]
            kv_rs, Nk, k_bs = 4 * C, kv_all.shape[0], 0
        ao3 = ws.get("x3_ao", (T, 3 * C), f16, dev)
        with profiling.region("global_attn_x3" if batch == 1 else "frame_attn_x3", (batch, tokens, Nk)):
            _C.flash_attn_x3(q, q_lo, k, k_lo, v, v_lo, ao3, C, batch, H, tokens, Nk, q_bs, q_rs, k_bs, kv_rs, k_bs, kv_rs,
                             tokens * 3 * C, 3 * C)
        with profiling.region("gemm", ("proj_x3", T, C, 3 * C)):
            _gemm_rows(ao3, pk["w3_proj"], x2d, bias=pk["b_proj"], gamma=pk["g1"], accumulate=True)
        _C.layernorm(x2d, pk["n2w"], pk["n2b"], xn3, self.norm2.eps, split3=True)
        h32 = ws.get("x3_h32", (T, Hd), torch.float32, dev)
        with profiling.region("gemm", ("fc1_x3", T, Hd, 3 * C)):
            _gemm_rows(xn3, pk["w3_fc1"], h32, bias=pk["b_fc1"])
        hid3 = ws.get("x3_hid", (T, 3 * Hd), f16, dev)
        _C.split3(h32, hid3, act=1)
        with profiling.region("gemm", ("fc2_x3", T, C, 3 * Hd)):
            _gemm_rows(hid3, pk["w3_fc2"], x2d, bias=pk["b_fc2"], gamma=pk["g2"], accumulate=True)
        return x2d

    def _attend_overlapped(self, qkv, kv_local, shard, qkmax, ao, ws, T, H, C, guard=None, guard_prev=None, overlap=True):
        """Multi-GPU global attention on the static-bound kernel, the K/V all-gather hidden behind the attention over this
        rank's own keys.  Partial results over disjoint key sets combine by a re-weighting that depends only on the shifts
        they were computed under (csrc/attention_v3.hip attn_combine_kernel), so the keys are processed as segments:
          1. the gather starts: K/V rows and, beside them, the 32 norm maxima the q/k-norm kernel just left -- every rank's key
             bound travels with its keys (64 bytes);
          2. own keys (from kv_local, while the gather is in flight; bound = this rank's maximum) -> slot world - 1;
          3. ONE launch over the gathered buffer in segment mode: one key range per rank, each under its OWN rank's bound, this
             rank's segment left out -- the same (world - 1) x query-tiles grid on every rank (two launches "ranks before /
             ranks after" cost a middle rank 6 rounds of workgroups where rank 0 needed 5) -> slots 0 .. world - 2;
          4. the combine kernel folds the slots and runs the flagged-tile fallback over all keys.
        overlap = False (IGGT_GATHER_OVERLAP=0): the gather completes first; same kernels."""
        W, r = shard.world, shard.rank
        dt, dev = qkv.dtype, qkv.device
        o_part = ws.get("attn_opart" + ("" if dt == precision.operand_dtype() else "_alt"), (W, 1, T, C), dt, dev)
        l_part = ws.get("attn_lpart", (W, 1, H, T), torch.float32, dev)
        c_part = ws.get("attn_cpart", (W, 1, H, T), torch.float32, dev)
        flags = ws.get("attn_flags", (H * ((T + 127) // 128),), torch.int32, dev)
        g = dict(guard=guard, guard_prev=guard_prev)
        with profiling.region("global_attn", (1, T, W * T)):
            if overlap:
                kv_all, stats_all, finish = shard.all_gather_kv_begin(kv_local, qkmax[:32])
            else:
                kv_all, stats_all = shard.all_gather_kv(kv_local, qkmax[:32])
                finish = None
            _C.flash_attn_d64_static_partial(qkv, kv_local, kv_local[:, C:], 1, H, T, T, 0, 3 * C, 0, 2 * C, 0, 2 * C, qkmax,
                                             o_part, l_part, c_part, W - 1, 1, **g)
            if finish is not None:
                finish()
            if W > 1:
                _C.flash_attn_d64_static_partial(qkv, kv_all, kv_all[:, C:], 1, H, T, W * T, 0, 3 * C, 0, 2 * C, 0, 2 * C,
                                                 None, o_part, l_part, c_part, 0, W, seg_len=T, skip_seg=r, seg_kmax=stats_all,
                                                 **g)
            _C.flash_attn_d64_static_combine(o_part, l_part, c_part, W, qkv, kv_all, kv_all[:, C:], ao, 1, H, T, W * T, 0,
                                             3 * C, 0, 2 * C, 0, 2 * C, 0, C, flags, **g)
        return True

    def attn_guard(self) -> Optional[torch.Tensor]:
        """This block's adaptive-switch word (None before the first forward / for blocks without q/k-norm)."""
        return None if self._packed is None else self._packed.get("guard")

    def forward(self, x: torch.Tensor, pos=None) -> torch.Tensor:
        """Reference signature (block.py:81): x [B, N, C] -> new tensor.  `pos` must be the standard
        aggregator layout (5 special tokens then a row-major grid) when the block has RoPE; it is only
        inspected on the host for its shape (no device sync)."""
        B, N, C = x.shape
        out = x.detach().float().contiguous().clone().view(B * N, C)
        rope_geom = None
        if self.attn.qk_norm and self.attn.rope is not None:
            raise _C.HipExtensionError(
                "RoPE blocks need the token-grid geometry: call Block.forward_inplace(..., rope_geom=...) "
                "or run them through Aggregator")
        self.forward_inplace(out, _default_ws(x.device), batch=B, tokens=N, rope_geom=rope_geom)
        return out.view(B, N, C)


NestedTensorBlock = Block  # reference block.py:210-259: Tensor input -> Block.forward

_WS = {}


def _default_ws(device) -> Workspace:
    key = str(device)
    if key not in _WS:
        _WS[key] = Workspace()
    return _WS[key]
