#!/usr/bin/env python3
"""One regime of probes/attn_static_robustness.py under the estimated-shift launch (forced), fp16, N = 43 968: rows handed over per
head, then 10 launches for a kernel trace (bash probes/profile_cmd.sh OUT probes/attn_est_regime.py <regime>)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import attn_static_robustness as r  # noqa: E402
from iggt_official_amd import _C  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "registers"
H, C, P, T = r.H, r.C, r.P, r.T
if kind.startswith("fewrows"):
    # LayerNorm-of-noise operands with the first `n` query rows at 30x norm: n x 16 rows end up on the lists, nothing else
    n = int(kind[7:] or 5)
    qkv, qkmax = r.make("noise", torch.float16)
    qkv.view(T, 3, H, 64)[:n, 0] *= 30.0
else:
    qkv, qkmax = r.make(kind, torch.float16)
o = torch.empty(T, C, dtype=torch.float16, device="cuda")
flags = torch.zeros(H * ((T + 127) // 128), dtype=torch.int32, device="cuda")
args = (qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, T, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C)
est_ws = torch.zeros(_C.static_attn_est_ws_bytes(1, H, T, T), dtype=torch.uint8, device="cuda")
for _ in range(10):
    _C.flash_attn_d64_static(*args, qkmax, flags, 0, None, None, None, est_ws=est_ws, key_period=P, key_nspecial=5, est_mode=1)
torch.cuda.synchronize()
v = _C.static_attn_est_views(est_ws, 1, H, T)
print(kind, "rows handed over per head:", v["rowcount"].tolist(), "capacity", v["NqL"])
