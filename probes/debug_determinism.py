#!/usr/bin/env python3
"""Does a forward at shape A give bit-identical results before and after a forward at a larger shape B?  Prints the first
stage that differs (DINOv2 tokens, aggregator layers, head outputs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import build_gpu_model  # noqa: E402
from oracle import weights  # noqa: E402

model = build_gpu_model("stress", 0)
a = weights.make_images(2, 56, 56, seed=1, device="cuda")
b = weights.make_images(3, 112, 84, seed=5, device="cuda")


def run(x):
    cap = {}
    h1 = model.aggregator.register_forward_hook(lambda m, i, o: cap.__setitem__("tokens", [t.clone() for t in o[0] if t is not None]))
    h2 = model.aggregator.patch_embed.register_forward_hook(lambda m, i, o: cap.__setitem__("dino", o["x_norm_patchtokens"].clone()))
    out = model(x)
    h1.remove(), h2.remove()
    torch.cuda.synchronize()
    d = {"dino": cap.get("dino")}
    for i, t in enumerate(cap["tokens"]):
        d[f"tokens{i}"] = t
    for k, v in out.items():
        if torch.is_tensor(v):
            d[k] = v.clone()
    d["pose"] = out["pose_enc"][-1].clone()
    return d


def cmp(x, y, tag):
    bad = [k for k in x if x[k] is not None and not torch.equal(x[k], y[k])]
    print(tag, "differs in:", bad if bad else "nothing")
    for k in bad[:3]:
        print("   ", k, float((x[k] - y[k]).abs().max()), float(x[k].abs().max()))


a1 = run(a)
a2 = run(a)
cmp(a1, a2, "A, A")
run(b)
a3 = run(a)
cmp(a1, a3, "A, B, A")
# what does B change?  try the suspects one at a time
model.aggregator._ws._bufs.clear()
a4 = run(a)
cmp(a1, a4, "A after clearing the aggregator workspace")
cmp(a3, a4, "   (vs A after B)")
