"""Host side of the implicit-GEMM convolution kernel (csrc/conv_igemm.hip): weight packing
(tap-major [Cout, KH*KW*Cin], bf16 hi/lo split, BatchNorm folding, transposed-conv re-indexing)
and launch helpers on NHWC fp32 activations.

Weight packs are cached per nn.Module and invalidated when the parameters change
(`load_state_dict`, `.to()`), like the GEMM packs of layers/blocks.py.
"""
from typing import List, Optional

import torch
import torch.nn as nn

from .. import _C, graphs, profiling

PREC = 3  # split-bf16 (fp32-grade) by default: the reference runs the heads in fp32 (vggt.py:189)


def _split(w: torch.Tensor):
    w = w.detach().float().contiguous()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


class PackedConv:
    __slots__ = ("w_hi", "w_lo", "bias", "KH", "KW", "stride", "pad_y", "pad_x", "Cin", "Cout", "cout_phys", "ps",
                 "osy", "osx", "ooy", "oox")

    def __init__(self, w2d, bias, KH, KW, stride, pad_y, pad_x, Cin, cout_phys=None, ps=1, osy=1, osx=1, ooy=0,
                 oox=0):
        self.w_hi, self.w_lo = _split(w2d)
        self.bias = None if bias is None else bias.detach().float().contiguous()
        self.KH, self.KW, self.stride, self.pad_y, self.pad_x = KH, KW, stride, pad_y, pad_x
        self.Cin, self.Cout = Cin, w2d.shape[0]
        self.cout_phys = self.Cout if cout_phys is None else cout_phys
        self.ps, self.osy, self.osx, self.ooy, self.oox = ps, osy, osx, ooy, oox


def _fold_bn(w, b, bn: Optional[nn.BatchNorm2d]):
    """Eval-mode BatchNorm folded into the preceding conv: w*(g/s), (b-mean)*(g/s)+beta."""
    if bn is None:
        return w, b
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    w = w.detach().float() * scale.view(-1, 1, 1, 1)
    b0 = torch.zeros_like(scale) if b is None else b.detach().float()
    return w, (b0 - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()


def pack_conv2d(conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d] = None, cin_pad: Optional[int] = None) -> PackedConv:
    w, b = _fold_bn(conv.weight, conv.bias, bn)
    w = w.detach().float()
    Cout, Cin, KH, KW = w.shape
    cin_p = cin_pad or Cin
    if cin_p % 32:
        raise ValueError(f"Cin {cin_p} must be a multiple of 32 (pad the channel stride)")
    if cin_p != Cin:
        w = torch.cat([w, w.new_zeros(Cout, cin_p - Cin, KH, KW)], 1)
    w2d = w.permute(0, 2, 3, 1).reshape(Cout, KH * KW * cin_p)
    assert conv.stride[0] == conv.stride[1] and conv.dilation == (1, 1) and conv.groups == 1
    return PackedConv(w2d, b, KH, KW, conv.stride[0], conv.padding[0], conv.padding[1], cin_p)


def pack_convT_kernel_eq_stride(ct: nn.ConvTranspose2d) -> PackedConv:
    """ConvTranspose2d(k = s, p = 0) == 1x1 GEMM to s*s*Cout channels + pixel shuffle."""
    w = ct.weight.detach().float()  # [Cin, Cout, s, s]
    Cin, Cout, s, s2 = w.shape
    assert s == s2 == ct.stride[0] == ct.stride[1] and ct.padding == (0, 0)
    w2d = w.permute(2, 3, 1, 0).reshape(s * s * Cout, Cin)
    bias = None if ct.bias is None else ct.bias.detach().float().repeat(s * s)
    return PackedConv(w2d, bias, 1, 1, 1, 0, 0, Cin, cout_phys=Cout, ps=s, osy=s, osx=s)


def pack_convT_k4s2p1(ct: nn.ConvTranspose2d) -> List[PackedConv]:
    """ConvTranspose2d(k4, s2, p1): output parity (py, px) is a 2x2 convolution of the input with taps
    ky = 3 - 2*jy (py = 0, pad 1) or 2 - 2*jy (py = 1, pad 0); four launches scatter with stride 2."""
    w = ct.weight.detach().float()  # [Cin, Cout, 4, 4]
    Cin, Cout = w.shape[:2]
    assert w.shape[2:] == (4, 4) and ct.stride == (2, 2) and ct.padding == (1, 1)
    packs = []
    for py in (0, 1):
        for px in (0, 1):
            kys = [3 - 2 * j for j in (0, 1)] if py == 0 else [2 - 2 * j for j in (0, 1)]
            kxs = [3 - 2 * j for j in (0, 1)] if px == 0 else [2 - 2 * j for j in (0, 1)]
            sub = w[:, :, kys][:, :, :, kxs]                      # [Cin, Cout, 2, 2] (jy, jx)
            w2d = sub.permute(1, 2, 3, 0).reshape(Cout, 4 * Cin)  # [Cout, (jy, jx, ci)]
            packs.append(PackedConv(w2d, ct.bias, 2, 2, 1, 1 - py, 1 - px, Cin, osy=2, osx=2, ooy=py, oox=px))
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
    with profiling.region("conv", 2.0 * N * Ho * Wo * pc.Cout * pc.KH * pc.KW * pc.Cin):
        _C.conv2d_nhwc(x, pc.w_hi, pc.w_lo, pc.bias, out, KH=pc.KH, KW=pc.KW, stride=pc.stride, pad_y=pc.pad_y,
                       pad_x=pc.pad_x, Ho=Ho, Wo=Wo, res=res, res2=res2, relu_in=relu_in, relu_res=relu_res, act=act,
                       prec=PREC if prec is None else prec, Cin=pc.Cin, Cout=pc.Cout, cout_phys=pc.cout_phys, ps=pc.ps,
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
