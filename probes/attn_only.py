"""Run only the global-attention kernel at the 32-view shape (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
P, C, H = 1374, 1024, 16
T = S * P
_C.load()
qkv = torch.randn(T, 3 * C, device="cuda").to(torch.bfloat16)
o = torch.empty(T, C, dtype=torch.bfloat16, device="cuda")
for _ in range(iters):
    _C.flash_attn_d64(qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, T, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C, 0.125, tile)
torch.cuda.synchronize()
print("done", float(o.float().abs().mean()))
