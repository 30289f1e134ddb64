"""Run only the global attention of ONE rank of a view-sharded run (layers/blocks.py Block._attend_overlapped: own keys, one
segment-mode launch over the gathered buffer, combine) at a BASELINE shape -- the target of rocprofv3 --pmc passes.
Usage: python probes/attn_rank_only.py [config4|config5] [f16|bf16] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C  # noqa: E402
from iggt_official_amd.layers.blocks import Block, Workspace  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "config4"
dt = torch.float16 if (len(sys.argv) <= 2 or sys.argv[2] == "f16") else torch.bfloat16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
H, C, W, r = 16, 1024, 8, 4
T = 4 * 1374 if cfg == "config4" else 8 * 5481
_C.load()
g = torch.Generator(device="cuda").manual_seed(1)
kv_all = torch.randn(W * T, 2 * C, generator=g, device="cuda").to(dt)
qkv = torch.randn(T, 3 * C, generator=g, device="cuda").to(dt)
qkv[:, :C] *= 0.125 * _C.LOG2E
kv_local = kv_all[r * T:(r + 1) * T].clone()
qkmax = torch.zeros(_C.QKMAX_NUMEL, device="cuda")
stats = torch.zeros(W, 32, device="cuda")
for s in range(W):
    _C.k_rownorm_max(kv_all[s * T:(s + 1) * T, :C], qkmax)
    stats[s] = qkmax[:32]
_C.k_rownorm_max(kv_local[:, :C], qkmax)


class Shard:
    world, rank = W, r

    def all_gather_kv_begin(self, kv, st):
        return kv_all, stats, (lambda: None)


ws = Workspace()
ao = torch.empty(T, C, dtype=dt, device="cuda")
for _ in range(iters):
    Block._attend_overlapped(None, qkv, kv_local, Shard(), qkmax, ao, ws, T, H, C)
torch.cuda.synchronize()
print("done", float(ao.float().abs().mean()))
