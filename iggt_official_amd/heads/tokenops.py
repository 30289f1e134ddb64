"""nn.LayerNorm / nn.Linear of the part head's token stages on HIP kernels (fp32 in, fp32 out).

The reference runs these in fp32 (autocast disabled for the heads, vggt.py:189).  LayerNorm -> `iggt_layernorm_f32`
(fp32 output); Linear -> the split-bf16 implicit-GEMM kernel used for the head convolutions, as a 1x1 convolution over
a [1, 1, tokens, C] NHWC view (fp32-grade: three MFMAs per product), with GELU / residual add fused into its epilogue.
Packs are cached per module and rebuilt when the parameters change."""
import weakref

import torch
import torch.nn as nn

from .. import _C
from . import convops as co

_PACKS = weakref.WeakKeyDictionary()
_LN_DIMS = (128, 256, 512, 1024, 2048)


def _need_cuda(x):
    if not x.is_cuda:
        raise _C.HipExtensionError("part-head token ops run on HIP kernels only (no CPU fallback)")


def layer_norm(norm: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
    """x [..., C] fp32 -> LayerNorm(x) fp32, C in {128, 256, 512, 1024, 2048}."""
    _need_cuda(x)
    C = x.shape[-1]
    if C not in _LN_DIMS:
        raise _C.HipExtensionError(f"HIP LayerNorm supports C in {_LN_DIMS}, got {C}")
    x2 = x.reshape(-1, C)
    if not x2.is_contiguous() or x2.dtype != torch.float32:
        x2 = x2.float().contiguous()
    out = torch.empty_like(x2)
    _C.layernorm(x2, norm.weight.detach().float(), norm.bias.detach().float(), out, norm.eps)
    return out.view(x.shape)


def linear(lin: nn.Linear, x: torch.Tensor, act: int = 0, res: torch.Tensor = None) -> torch.Tensor:
    """act(x @ W^T + b) (+ res); act 0 none / 1 ReLU / 3 exact GELU; x [..., Cin] fp32 with Cin % 32 == 0."""
    _need_cuda(x)
    Cin, Cout = lin.in_features, lin.out_features
    if Cin % 32:
        raise _C.HipExtensionError(f"HIP Linear needs in_features % 32 == 0, got {Cin}")
    cache = _PACKS.get(lin)
    if cache is None:
        cache = _PACKS[lin] = co.PackCache()
    pc = cache.get(0, (lin.weight, lin.bias),
                   lambda: co.PackedConv(lin.weight.detach().float(), lin.bias, 1, 1, 1, 0, 0, Cin))
    x2 = x.reshape(1, 1, -1, Cin)
    if not x2.is_contiguous() or x2.dtype != torch.float32:
        x2 = x2.float().contiguous()
    r2 = None
    if res is not None:
        r2 = res.reshape(1, 1, -1, Cout)
        if not r2.is_contiguous() or r2.dtype != torch.float32:
            r2 = r2.float().contiguous()
    y = co.run(pc, x2, act=act, res=r2, prec=co.PART_PREC)
    return y.view(*x.shape[:-1], Cout)
