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


# ---------------------------------------------------------------------------------------------------------------------------
# Round 6: the MLPs of the part head's window-attention stages on the trunk's 16-bit path.  At 32 views @ 532^2 the HAB stage
# holds 2.96 M tokens of 128 channels: as 1 x 1 convolutions on fp32 maps its MLP read 1.5 GB and wrote a 6 GB fp32 hidden map, read
# it again and wrote 1.5 GB -- HBM-shaped work on a kernel built for K in the thousands (~100 TFLOP/s).  Here, exactly like a trunk
# block (layers/blocks.py): LayerNorm -> fp16, fc1 + exact GELU (LDS table) -> fp16 hidden on the 256^2 LDS-DMA GEMM, fc2 +
# residual accumulated IN PLACE into the fp32 token matrix, fp16 weights with mean-input compensation.  Active with the two-pass
# part branch (convops.PART_PREC == 2; IGGT_PART_CONV_PREC=3 keeps every Linear fp32-grade); part_feat stays within its budget
# (profiles/r06_part_branch_ab.txt).
_H16_MIN_TOKENS = 8192
_WS = {}


def _ws(device):
    from ..layers.blocks import Workspace

    key = str(device)          # the part branch runs on the caller's stream only (models/vggt.py)
    if key not in _WS:
        _WS[key] = Workspace()
    return _WS[key]


def mlp_h16_applicable(fc1: nn.Linear, fc2: nn.Linear, tokens: int) -> bool:
    from .. import precision

    return (co.PART_PREC == 2 and precision.operand_dtype() == torch.float16 and tokens >= _H16_MIN_TOKENS
            and fc1.in_features % 64 == 0 and fc1.out_features % 256 == 0 and fc2.out_features % 128 == 0
            and fc2.in_features == fc1.out_features and fc2.out_features == fc1.in_features)


def mlp_h16_(norm: nn.LayerNorm, fc1: nn.Linear, fc2: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """x [..., C] fp32 contiguous, updated IN PLACE: x += fc2(GELU(fc1(LayerNorm(x)))) (reference window_sa.py:225,317; mlp 83-99)."""
    from ..layers.blocks import _gemm_rows, _h16_residual, _h16_weight, compensated_bias

    _need_cuda(x)
    C, Hd = fc1.in_features, fc1.out_features
    x2 = x.view(-1, C)
    assert x2.is_contiguous() and x2.dtype == torch.float32
    T = x2.shape[0]
    cache = _PACKS.get(fc1)
    if cache is None:
        cache = _PACKS[fc1] = co.PackCache()
    f16 = torch.float16

    def pack():
        w1, w2 = fc1.weight.detach().float(), fc2.weight.detach().float()
        if not (float(torch.stack([w1.abs().amax(), w2.abs().amax()]).amax()) <= 65504.0):
            raise _C.HipExtensionError("part-head MLP weights exceed the fp16 range: run with IGGT_PART_CONV_PREC=3")
        return (_h16_weight(w1, f16), _h16_residual(w1, f16), fc1.bias.detach().float().contiguous(),
                _h16_weight(w2, f16), _h16_residual(w2, f16), fc2.bias.detach().float().contiguous())

    w1h, dw1, b1, w2h, dw2, b2 = cache.get("mlp_h16", (fc1.weight, fc1.bias, fc2.weight, fc2.bias), pack)
    ws = _ws(x.device)
    xn = ws.get("xn16", (T, C), f16, x.device)
    _C.layernorm(x2, norm.weight.detach().float(), norm.bias.detach().float(), xn, norm.eps)
    hid = ws.get("hid16", (T, Hd), f16, x.device)
    _gemm_rows(xn, w1h, hid, bias=compensated_bias(ws, xn, dw1, b1), act=1)
    _gemm_rows(hid, w2h, x2, bias=compensated_bias(ws, hid, dw2, b2), accumulate=True)
    return x
