"""Run only the part head's two window-attention launches at the 32-view @ 532^2 shapes (HAB: 304^2 x 128 channels, 4 heads x 32;
OCAB: 152^2 x 256 channels, 4 heads x 64, 12 x 12 key windows, bias) -- the target of rocprofv3 --pmc passes.
Usage: python probes/window_attn_only.py [frames] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C  # noqa: E402

_C.load()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator(device="cuda").manual_seed(0)
h = w = 304
qkv = torch.randn(b, h, w, 384, generator=g, device="cuda")
o = torch.empty(b, h, w, 128, device="cuda")
h2 = w2 = 152
nW = b * (h2 // 8) * (w2 // 8)
qw = torch.randn(nW, 64, 256, generator=g, device="cuda")
kk, vv = torch.randn(b, h2, w2, 256, generator=g, device="cuda"), torch.randn(b, h2, w2, 256, generator=g, device="cuda")
bias = torch.randn(4, 144, 64, generator=g, device="cuda") * 0.5
o2 = torch.empty(b, h2, w2, 256, device="cuda")
for _ in range(iters):
    _C.window_attn(qkv[..., :128], qkv[..., 128:256], qkv[..., 256:], o, 4, 32, 32 ** -0.5)
    _C.window_attn(qw, kk, vv, o2, 4, 64, 64 ** -0.5, q_windows=True, ow=12, pad=2, bias=bias)
torch.cuda.synchronize()
print("ok", float(o.abs().mean()), float(o2.abs().mean()))
