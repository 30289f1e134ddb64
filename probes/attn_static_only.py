"""Run only the production global-attention launch (static-bound kernel + gated pass, or the online-max kernel with
arg 'online') at the 32-view bench shape -- the target of the rocprofv3 --pmc passes in probes/pmc_attn.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "static"
dt = torch.float16 if (len(sys.argv) <= 2 or sys.argv[2] == "f16") else torch.bfloat16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
S, P, C, H = 32, 1374, 1024, 16
T = S * P
_C.load()
g = torch.Generator(device="cpu").manual_seed(1)
qkv = torch.randn(T, 3 * C, generator=g).to(dt).cuda()
o = torch.empty(T, C, dtype=dt, device="cuda")
if mode == "static":
    qkv[:, :C] *= 0.125 * _C.LOG2E
    x = qkv.view(T, 3, H, 64)
    qkmax = torch.zeros(32, device="cuda")
    qkmax[:16] = x[:, 0].float().norm(dim=-1).amax(0)
    qkmax[16:] = x[:, 1].float().norm(dim=-1).amax(0)
    flags = torch.zeros(H * ((T + 127) // 128), dtype=torch.int32, device="cuda")
for _ in range(iters):
    if mode == "static":
        _C.flash_attn_d64_static(qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, T, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C, qkmax,
                                 flags, 0)
    else:
        _C.flash_attn_d64(qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, T, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C, 0.125, 0)
torch.cuda.synchronize()
print("done", float(o.float().abs().mean()))
