#!/usr/bin/env python3
"""Why does a RE-captured graph of shape A differ (in the last digits) from the eager forward?  Variants that isolate the cause."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import build_gpu_model  # noqa: E402
from iggt_official_amd import graphs, precision  # noqa: E402
from oracle import weights  # noqa: E402

model = build_gpu_model("stress", 0)
a = weights.make_images(2, 56, 56, seed=1, device="cuda")
b = weights.make_images(3, 112, 84, seed=5, device="cuda")
KEYS = ("depth", "depth_conf", "world_points", "world_points_conf", "part_feat")


def copy(p):
    d = {k: p[k].clone() for k in KEYS}
    d["pose"] = p["pose_enc"][-1].clone()
    return d


def cmp(x, y, tag):
    bad = [(k, float((x[k] - y[k]).abs().max() / y[k].abs().max())) for k in x if not torch.equal(x[k], y[k])]
    print(f"{tag:70s}", bad if bad else "identical", flush=True)


eager = copy(model(a))
model.enable_graphs(True)
cmp(copy(model(a)), eager, "graph #1 (fresh)")
graphs.buffers_changed()
cmp(copy(model(a)), eager, "re-capture forced by buffers_changed() alone")
cmp(copy(model(a)), eager, "   replay of it")
model.enable_graphs(False)
model.enable_graphs(True)
cmp(copy(model(a)), eager, "graphs off/on -> fresh capture")
model(b)
cmp(copy(model(a)), eager, "after capturing B: re-capture of A")
model.enable_graphs(False)
cmp(copy(model(a)), eager, "eager again")
for name, fn in (("mean compensation off", lambda: precision.set_mean_compensation(False)),
                 ("static softmax off", lambda: precision.set_static_softmax(False))):
    fn()
    e2 = copy(model(a))
    model.enable_graphs(True)
    cmp(copy(model(a)), e2, f"[{name}] graph #1")
    model(b)
    cmp(copy(model(a)), e2, f"[{name}] after capturing B: re-capture of A")
    model.enable_graphs(False)
    precision.set_mean_compensation(True)
    precision.set_static_softmax(True)
