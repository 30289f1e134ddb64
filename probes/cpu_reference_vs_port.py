#!/usr/bin/env python3
"""How far is bench.py's `cpu_baseline` (kind "port": oracle/restate.py) from the REFERENCE's own CPU path?  BUILD CONTAINER ONLY
(needs /root/reference; a Python reference cannot travel to the GPU box in any form, so the bench line there can only time the
port -- this probe measures, on ONE set of cores and with ONE protocol, what that substitution is worth).

Protocol: S views @ 518 x 518 (default 4 = BASELINE.json configs[0]'s size), geometry outputs (aggregator +
camera / depth / point heads; `part_feat` is undefined in the reference at 518), fp32, torch.set_num_threads(all cores), the same
seeded synthetic weights and images on both sides, ONE warm-up forward of the same size, then one timed forward
(time.perf_counter).  Reference: the reference's own modules through oracle/ref_shim.py, heads called with frames_chunk_size=None.
Usage: python probes/cpu_reference_vs_port.py [S] > profiles/r06_cpu_reference_vs_port.txt"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, restate, weights  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H = 518
cores = len(os.sched_getaffinity(0))
torch.set_num_threads(cores)
with open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")) as f:
    schema = json.load(f)
sd = weights.fill_state_dict(schema, seed=0, mode="default")
images = weights.make_images(S, H, H, seed=0)
cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
print(f"host: {cores} usable cores, {cpu[0] if cpu else '?'}; torch {torch.__version__}, {torch.get_num_threads()} threads; "
      f"{S} views @ {H}x{H}, fp32, geometry outputs")


def timed(fn):
    fn()                                   # warm-up: same size (thread pool, allocator, oneDNN primitive caches)
    t0 = time.perf_counter()
    out = fn()
    return time.perf_counter() - t0, out


# ---- the port (what bench.py times on the GPU box) -----------------------------------------------------------------------------
with torch.no_grad():
    dt_port, out_p = timed(lambda: restate.iggt_forward(sd, images, with_part=False))
print(f"port      (oracle/restate.py iggt_forward):                 {dt_port:8.2f} s  = {S / dt_port:.4f} views/s")

# ---- the reference's own modules -----------------------------------------------------------------------------------------------
model = ref_shim.build_reference_iggt(fast_init=True)
missing, unexpected = model.load_state_dict(sd, strict=False)
assert not unexpected
im5 = images[None]


def ref_forward():
    tokens, psi = model.aggregator(im5)
    pose = model.camera_head(tokens)
    depth, dconf = model.depth_head(tokens, images=im5, patch_start_idx=psi, frames_chunk_size=None)
    pts, pconf, _ = model.point_head(tokens, images=im5, patch_start_idx=psi, frames_chunk_size=None)
    return dict(depth=depth, world_points=pts, pose=pose[-1])


with torch.no_grad():
    dt_ref, out_r = timed(ref_forward)
print(f"reference (iggt.models / iggt.heads modules through the shim): {dt_ref:8.2f} s  = {S / dt_ref:.4f} views/s")
print(f"port / reference speed ratio on the same cores: {dt_ref / dt_port:.2f}x")
for k, a, b in (("depth", out_p["depth"], out_r["depth"]), ("world_points", out_p["world_points"], out_r["world_points"])):
    print(f"  same outputs: {k} relative l2 {float((a.double() - b.double()).norm() / b.double().norm()):.2e}")
print("what differs: the reference keeps all 24 [frame | global] concatenations (24 x S x P x 2048 floats) and runs RoPE through\n"
      "per-call position tables with a device->host sync (rope.py:177); the port keeps 4 layers.  Same ATen kernels otherwise.")
