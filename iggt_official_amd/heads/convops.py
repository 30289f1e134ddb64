"""Host side of the implicit-GEMM convolution kernel (csrc/conv_igemm.hip): weight packing
(tap-major [Cout, KH*KW*Cin], bf16 hi/lo split, BatchNorm folding, transposed-conv re-indexing)
and launch helpers on NHWC fp32 activations.

Weight packs are cached per nn.Module and invalidated when the parameters change
(`load_state_dict`, `.to()`), like the GEMM packs of layers/blocks.py.
"""
import os
from typing import List, Optional

import torch
import torch.nn as nn

from .. import _C, graphs, profiling

PREC = 3  # split-bf16 (fp32-grade) by default: the reference runs the heads in fp32 (vggt.py:189)


def _env_prec(name: str, default: int) -> int:
    v = os.environ.get(name, "").strip()
    if v == "":
        return default
    if v not in ("2", "3"):
        raise ValueError(f"{name} must be 2 or 3 (got {v!r})")
    return int(v)


# The DPT depth / point heads (heads/dpt_head.py) run their convolutions at this precision.  2 = fp16 activations hi + lo
# against weights rounded once to fp16, with the mean-input compensation of csrc/conv_meancomp.hip: two MFMA passes instead of
# three.  Measured per layer on photographs in profiles/r03_conv_precision.txt (head output error 1.7e-4 with every layer at
# 2, against 1.2e-5 at 3 and a trunk contribution of 3.8e-4 / 7.2e-4).  IGGT_CONV_PREC=3 restores the split-bf16 kernels.
DPT_PREC = _env_prec("IGGT_CONV_PREC", 2)
# The instance-feature branch (heads/adaptor.py SamProjector, heads/part_head.py, heads/window_sa.py, heads/tokenops.py): its 3 x 3
# convolutions and token Linears at this precision (same two-pass arithmetic and the same size rule as DPT_PREC when 2).  Measured
# at 32 x 532^2 against the reference fixture (round 6, profiles/r06_part_branch_ab.txt): part_feat 1.48e-4 at 3, 1.7e-4 at 2.
PART_PREC = _env_prec("IGGT_PART_CONV_PREC", 2)
# ... for convolutions of at least this many FLOPs (2 * pixels * Cout * taps * Cin).  Below it the channel-mean and correction
# launches in front of a prec-2 convolution (~15 us) outweigh the MFMA pass they save: measured break-even between the 37^2 and
# the 74^2 maps of a 32-view pass (profiles/r03_conv_prec_ab.txt); the layer then runs at prec 3, which is the more accurate one.
PREC2_MIN_FLOPS = 1.0e11


def _split(w: torch.Tensor):
    w = w.detach().float().contiguous()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


def _split_f16(w: torch.Tensor):
    """prec 2: (fp16(w) -- the MFMA operand, saturated to the finite fp16 range; bf16(w - fp16(w)) -- only ever multiplied by
    the channel means, conv_meancomp.hip; bf16 keeps the RANGE of a 2^-12 |w| residual, which fp16 would flush)."""
    w = w.detach().float().contiguous()
    h = w.clamp(-65504.0, 65504.0).to(torch.float16)
    return h.contiguous(), (w - h.float()).to(torch.bfloat16).contiguous()


class PackedConv:
    __slots__ = ("_src", "_planes", "bias", "KH", "KW", "stride", "pad_y", "pad_x", "Cin", "Cout", "cout_phys", "ps",
                 "osy", "osx", "ooy", "oox")

    def __init__(self, w2d, bias, KH, KW, stride, pad_y, pad_x, Cin, cout_phys=None, ps=1, osy=1, osx=1, ooy=0,
                 oox=0, Cout=None):
        """w2d: the fp32 weights [Cout, KH * KW * Cin] (tap-major), or a zero-argument callable that returns them (then `Cout`
        must be given).  The operand planes are built from it on demand, per precision -- always from the fp32 values, so a
        layer's planes do not depend on which precision it happened to run first; the pack functions below pass a closure over
        the nn.Module, which keeps no fp32 copy alive (the PackCache drops the pack when the module's tensors change)."""
        self._src = w2d if callable(w2d) else w2d.detach()
        self._planes = {}
        self.bias = None if bias is None else bias.detach().float().contiguous()
        self.KH, self.KW, self.stride, self.pad_y, self.pad_x = KH, KW, stride, pad_y, pad_x
        self.Cin, self.Cout = Cin, (w2d.shape[0] if Cout is None else Cout)
        self.cout_phys = self.Cout if cout_phys is None else cout_phys
        self.ps, self.osy, self.osx, self.ooy, self.oox = ps, osy, osx, ooy, oox

    def planes(self, prec: int):
        """The two weight tensors the kernels take at `prec`: 3 (and 1) -> bf16 hi, bf16 lo; 2 -> fp16, bf16 residual."""
        key = 2 if prec == 2 else 3
        pl = self._planes.get(key)
        if pl is None:
            w = self._src() if callable(self._src) else self._src
            w = w.detach().float().contiguous()
            assert w.dim() == 2 and w.shape[0] == self.Cout and w.shape[1] == self.KH * self.KW * self.Cin, w.shape
            pl = self._planes[key] = _split_f16(w) if key == 2 else _split(w)
        return pl

    @property
    def w_hi(self):
        return self.planes(3)[0]

    @property
    def w_lo(self):
        return self.planes(3)[1]


def _fold_bn(w, b, bn: Optional[nn.BatchNorm2d]):
    """Eval-mode BatchNorm folded into the preceding conv: w*(g/s), (b-mean)*(g/s)+beta."""
    if bn is None:
        return w, b
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    w = w.detach().float() * scale.view(-1, 1, 1, 1)
    b0 = torch.zeros_like(scale) if b is None else b.detach().float()
    return w, (b0 - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()


def pack_conv2d(conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d] = None, cin_pad: Optional[int] = None) -> PackedConv:
    Cout, Cin, KH, KW = conv.weight.shape
    cin_p = cin_pad or Cin
    if cin_p % 32:
        raise ValueError(f"Cin {cin_p} must be a multiple of 32 (pad the channel stride)")
    assert conv.stride[0] == conv.stride[1] and conv.dilation == (1, 1) and conv.groups == 1

    def w2d():
        w = _fold_bn(conv.weight, conv.bias, bn)[0].detach().float()
        if cin_p != Cin:
            w = torch.cat([w, w.new_zeros(Cout, cin_p - Cin, KH, KW)], 1)
        return w.permute(0, 2, 3, 1).reshape(Cout, KH * KW * cin_p)

    b = _fold_bn(conv.weight, conv.bias, bn)[1]
    return PackedConv(w2d, b, KH, KW, conv.stride[0], conv.padding[0], conv.padding[1], cin_p, Cout=Cout)


def pack_convT_kernel_eq_stride(ct: nn.ConvTranspose2d) -> PackedConv:
    """ConvTranspose2d(k = s, p = 0) == 1x1 GEMM to s*s*Cout channels + pixel shuffle."""
    Cin, Cout, s, s2 = ct.weight.shape    # [Cin, Cout, s, s]
    assert s == s2 == ct.stride[0] == ct.stride[1] and ct.padding == (0, 0)
    bias = None if ct.bias is None else ct.bias.detach().float().repeat(s * s)
    return PackedConv(lambda: ct.weight.detach().float().permute(2, 3, 1, 0).reshape(s * s * Cout, Cin), bias, 1, 1, 1, 0, 0,
                      Cin, cout_phys=Cout, ps=s, osy=s, osx=s, Cout=s * s * Cout)


def pack_convT_k4s2p1(ct: nn.ConvTranspose2d) -> List[PackedConv]:
    """ConvTranspose2d(k4, s2, p1): output parity (py, px) is a 2x2 convolution of the input with taps
    ky = 3 - 2*jy (py = 0, pad 1) or 2 - 2*jy (py = 1, pad 0); four launches scatter with stride 2."""
    Cin, Cout = ct.weight.shape[:2]       # [Cin, Cout, 4, 4]
    assert ct.weight.shape[2:] == (4, 4) and ct.stride == (2, 2) and ct.padding == (1, 1)
    packs = []
    for py in (0, 1):
        for px in (0, 1):
            kys = [3 - 2 * j for j in (0, 1)] if py == 0 else [2 - 2 * j for j in (0, 1)]
            kxs = [3 - 2 * j for j in (0, 1)] if px == 0 else [2 - 2 * j for j in (0, 1)]

            def w2d(kys=kys, kxs=kxs):
                sub = ct.weight.detach().float()[:, :, kys][:, :, :, kxs]   # [Cin, Cout, 2, 2] (jy, jx)
                return sub.permute(1, 2, 3, 0).reshape(Cout, 4 * Cin)       # [Cout, (jy, jx, ci)]

            packs.append(PackedConv(w2d, ct.bias, 2, 2, 1, 1 - py, 1 - px, Cin, osy=2, osx=2, ooy=py, oox=px, Cout=Cout))
    return packs


class PackCache:
    """module -> pack, rebuilt when any of the module's tensors changed."""

    def __init__(self):
        self._d = {}

    def get(self, key, tensors, build):
        ver = tuple((t.data_ptr(), t._version) for t in tensors if t is not None)
        hit = self._d.get(key)
        if hit is None or hit[0] != ver:
            if hit is not None:
                graphs.buffers_changed()    # the old pack is freed here; captured graphs hold its address
            self._d[key] = (ver, build())
        return self._d[key][1]


def run(pc: PackedConv, x: torch.Tensor, *, out: Optional[torch.Tensor] = None, relu_in=False, res=None,
        relu_res=False, res2=None, act=0, prec: Optional[int] = None, ldy: Optional[int] = None) -> torch.Tensor:
    """x NHWC fp32 [N,Hi,Wi,C>=Cin] -> NHWC fp32.  Output size from the conv geometry."""
    N, Hi, Wi, ldx = x.shape
    if pc.ps > 1:                                  # kernel == stride transposed conv
        Ho, Wo = Hi, Wi
        Hout, Wout = Hi * pc.ps, Wi * pc.ps
    elif pc.osy > 1:                               # parity sub-conv of a k4 s2 p1 transposed conv
        Ho, Wo = Hi, Wi
        Hout, Wout = Hi * 2, Wi * 2
    else:
        Ho = (Hi + 2 * pc.pad_y - pc.KH) // pc.stride + 1
        Wo = (Wi + 2 * pc.pad_x - pc.KW) // pc.stride + 1
        Hout, Wout = Ho, Wo
    if out is None:
        c = pc.cout_phys if ldy is None else ldy
        out = (torch.zeros if (ldy and ldy != pc.cout_phys) else torch.empty)(
            N, Hout, Wout, c, dtype=torch.float32, device=x.device)
    # bench.py's secondary roofline leg: algorithmic FLOPs of this convolution = 2 * output pixels * Cout * taps * Cin
    prec = PREC if prec is None else prec
    flops = 2.0 * N * Ho * Wo * pc.Cout * pc.KH * pc.KW * pc.Cin
    if prec == 2 and flops < PREC2_MIN_FLOPS:
        prec = 3          # small problem: the two correction launches cost more than the third MFMA pass
    if prec == 2 and (max(pc.pad_y, pc.pad_x) > pc.stride or pc.KH * pc.KW > 16 or pc.Cin > 8192 or pc.Cin % 8):
        # geometries whose mean-input compensation has no nine-border-class description (csrc/conv_igemm.hip returns -6 for
        # them at two passes): decided here, up front, so that a captured graph always holds the same kernels.  The stock DPT
        # configuration never gets here; custom heads / strides may.
        prec = 3
    w_a, w_b = pc.planes(prec)
    # (algorithmic FLOPs, MFMA passes per product, geometry): bench.py's rooflines read the first two, probes/part_branch_table.py all
    with profiling.region("conv", (flops, 2 if prec == 2 else 3, (N, Ho, Wo, pc.Cin, pc.Cout, pc.KH, pc.stride))):
        _C.conv2d_nhwc(x, w_a, w_b, pc.bias, out, KH=pc.KH, KW=pc.KW, stride=pc.stride, pad_y=pc.pad_y,
                       pad_x=pc.pad_x, Ho=Ho, Wo=Wo, res=res, res2=res2, relu_in=relu_in, relu_res=relu_res, act=act,
                       prec=prec, Cin=pc.Cin, Cout=pc.Cout, cout_phys=pc.cout_phys, ps=pc.ps,
                       osy=pc.osy, osx=pc.osx, ooy=pc.ooy, oox=pc.oox)
    return out


def run_convT_k4s2p1(packs: List[PackedConv], x: torch.Tensor, prec: Optional[int] = None) -> torch.Tensor:
    N, Hi, Wi, _ = x.shape
    out = torch.empty(N, 2 * Hi, 2 * Wi, packs[0].Cout, dtype=torch.float32, device=x.device)
    for pc in packs:
        run(pc, x, out=out, prec=prec)
    return out


def resize(x: torch.Tensor, size, xpart=None, ypart=None) -> torch.Tensor:
    """Bilinear align_corners=True resize of NHWC fp32 (+ optional separable position map)."""
    N, Hi, Wi, C = x.shape
    out = torch.empty(N, size[0], size[1], C, dtype=torch.float32, device=x.device)
    return _C.bilinear_ac_nhwc(x.contiguous(), out, xpart, ypart)
