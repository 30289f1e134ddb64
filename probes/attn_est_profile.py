#!/usr/bin/env python3
"""Per-kernel cost of the estimated-shift launch (run under rocprofv3 --kernel-trace --stats): 10 launches each of the norm-bound
static kernel, the estimated-shift launch (forced) on LayerNorm-of-noise and on sink-key operands, fp16, N = 43 968."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import attn_static_robustness as r  # noqa: E402
from iggt_official_amd import _C  # noqa: E402

H, C, P, T = r.H, r.C, r.P, r.T
for kind in ("noise", "sinks", "registers"):
    qkv, qkmax = r.make(kind, torch.float16)
    o = torch.empty(T, C, dtype=torch.float16, device="cuda")
    flags = torch.zeros(H * ((T + 127) // 128), dtype=torch.int32, device="cuda")
    args = (qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, T, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C)
    est_ws = torch.zeros(_C.static_attn_est_ws_bytes(1, H, T, T), dtype=torch.uint8, device="cuda")
    for _ in range(10):
        _C.flash_attn_d64_static(*args, qkmax, flags, 0, None, None, None, est_ws=est_ws, key_period=P, key_nspecial=5, est_mode=1)
    torch.cuda.synchronize()
print("done")
