#!/usr/bin/env python3
"""Kernel table of the instance-feature branch ALONE (SamProjector + PartHead) inside a whole forward at S views @ HxW: the torch
profiler is switched on in a pre-hook of part_adaptor and off in a hook of part_head (device synchronised at both ends).
Usage: python probes/part_kernels.py [S H W] > profiles/r06_part_kernels.txt   (GPU box)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from iggt.models.vggt import IGGT  # noqa: E402
from iggt_official_amd import synthetic  # noqa: E402

a = sys.argv[1:]
S, H, W = (int(x) for x in (a[:3] if len(a) >= 3 else (32, 532, 532)))
with torch.device("cuda"):
    model = IGGT().eval()
with open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")) as f:
    model.load_state_dict(synthetic.fill_state_dict(json.load(f), seed=0, mode="stress", device="cuda"), strict=False)
img = synthetic.make_images(S, H, W, seed=11, device="cuda")
for _ in range(2):
    model(img)
torch.cuda.synchronize()
prof = profile(activities=[ProfilerActivity.CUDA])


def start(mod, args, kwargs=None):
    torch.cuda.synchronize()
    prof.__enter__()


def stop(mod, args, out):
    torch.cuda.synchronize()
    prof.__exit__(None, None, None)


h0 = model.part_adaptor.register_forward_pre_hook(start)
h1 = model.part_head.register_forward_hook(stop)
model(img)
torch.cuda.synchronize()
h0.remove(); h1.remove()
rows = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        d = rows.setdefault(e.name, [0, 0.0, 0.0])
        t = e.device_time if hasattr(e, "device_time") else e.cuda_time
        d[0] += 1; d[1] += t; d[2] = max(d[2], t)
tot = sum(d[1] for d in rows.values())
print(f"part branch kernels, {S} views @ {H}x{W}: {tot / 1e3:.2f} ms of kernel time in {sum(d[0] for d in rows.values())} launches")
print(f"{'kernel':120s} {'calls':>5s} {'total_ms':>9s} {'avg_us':>9s} {'max_us':>9s} {'%':>6s}")
for k, d in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:120]:120s} {d[0]:5d} {d[1] / 1e3:9.3f} {d[1] / d[0]:9.1f} {d[2]:9.1f} {100 * d[1] / tot:6.2f}")
