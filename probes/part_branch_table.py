#!/usr/bin/env python3
"""Per-launch table of the instance-feature branch (SamProjector + PartHead) at S views @ HxW: every convolution / Linear launch with
its geometry, MFMA passes and HIP-event time, the window / cross attention launches, and what is left (LayerNorms, resizes, the
torch element-wise glue).  Usage: python probes/part_branch_table.py [S H W] > profiles/r06_part_branch_table.txt  (GPU box)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from iggt.models.vggt import IGGT  # noqa: E402
from iggt_official_amd import profiling, synthetic  # noqa: E402
from iggt_official_amd.models import vggt as _mv  # noqa: E402

a = sys.argv[1:]
S, H, W = (int(x) for x in (a[:3] if len(a) >= 3 else (32, 532, 532)))
with torch.device("cuda"):
    model = IGGT(part_on_invalid_grid="skip").eval()      # 518: geometry outputs only -> the DPT table alone
with open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")) as f:
    model.load_state_dict(synthetic.fill_state_dict(json.load(f), seed=0, mode="stress", device="cuda"), strict=False)
img = synthetic.make_images(S, H, W, seed=11, device="cuda")
for _ in range(2):
    model(img)
torch.cuda.synchronize()
_mv._HEAD_STREAMS = "0"
names = ("conv", "window_attn", "cross_attn", "part_branch")
for n in names:
    profiling.enable(n)
model(img)
torch.cuda.synchronize()
recs = {n: profiling.summarize(profiling.disable(n)) for n in names}
part = [r for r in recs["conv"] if r[2] == "part"]
total = sum(r[0] for r in recs["part_branch"]) or 1e-9
print(f"part branch, {S} views @ {H}x{W}: {total:.2f} ms (HIP events around part_adaptor + part_head, heads in line)")
print(f"{'launch':>6s} {'N':>3s} {'Ho':>4s} {'Wo':>4s} {'Cin':>5s} {'Cout':>5s} {'k':>2s} {'s':>2s} {'passes':>6s} {'GFLOP':>9s} {'ms':>8s} {'TFLOP/s':>8s}")
agg = {}
for i, (ms, (fl, ps, geo), _t) in enumerate(part):
    print(f"{i:6d} {geo[0]:3d} {geo[1]:4d} {geo[2]:4d} {geo[3]:5d} {geo[4]:5d} {geo[5]:2d} {geo[6]:2d} {ps:6d} {fl / 1e9:9.1f} {ms:8.3f} {fl / ms / 1e9:8.1f}")
    k = (geo[1], geo[3], geo[4], geo[5], ps)
    d = agg.setdefault(k, [0, 0.0, 0.0])
    d[0] += 1; d[1] += ms; d[2] += fl
print("\nby (map height, Cin, Cout, kernel, passes), largest first:")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k[0]:4d}^2 {k[1]:5d} -> {k[2]:5d}  k{k[3]}  passes {k[4]}: {d[0]:3d} launches {d[1]:8.3f} ms  {d[2] / d[1] / 1e9:7.1f} TFLOP/s")
dpt = [r for r in recs["conv"] if r[2] != "part"]
agg2 = {}
for ms, (fl, ps, geo), _t in dpt:
    k = (geo[1], geo[3], geo[4], geo[5], geo[6], ps)
    d = agg2.setdefault(k, [0, 0.0, 0.0])
    d[0] += 1; d[1] += ms; d[2] += fl
print(f"\nthe two DPT heads at this size, {sum(r[0] for r in dpt):.2f} ms in {len(dpt)} launches, by (map height, Cin, Cout, kernel, stride, passes):")
for k, d in sorted(agg2.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k[0]:4d}^2 {k[1]:5d} -> {k[2]:5d}  k{k[3]} s{k[4]}  passes {k[5]}: {d[0]:3d} launches {d[1]:8.3f} ms  {d[2] / d[1] / 1e9:7.1f} TFLOP/s")
conv_ms = sum(r[0] for r in part)
wa = sum(r[0] for r in recs["window_attn"]); ca = sum(r[0] for r in recs["cross_attn"])
print(f"\nconvolutions + Linears {conv_ms:.2f} ms ({sum(r[1][0] for r in part) / 1e12:.2f} TFLOP), window attention {wa:.2f} ms, cross attention {ca:.2f} ms, "
      f"everything else (LayerNorms, resizes, token projection of the SamProjector, element-wise glue) {total - conv_ms - wa - ca:.2f} ms")
