#!/usr/bin/env python3
"""Does any kernel of the forward read memory it did not write?  Forward at shape A, then poison the caching allocator's free
blocks (NaN / zeros / 1e4) and run A again: every output must be bit-identical.  Prints the first stages that differ."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import build_gpu_model  # noqa: E402
from oracle import weights  # noqa: E402

model = build_gpu_model("stress", 0)
shape = tuple(int(x) for x in (sys.argv[1:4] or (2, 56, 56)))
a = weights.make_images(*shape, seed=1, device="cuda")


def run(x):
    cap = {}
    h1 = model.aggregator.register_forward_hook(lambda m, i, o: cap.__setitem__("tokens", {i: t.clone() for i, t in enumerate(o[0]) if t is not None}))
    out = model(x)
    h1.remove()
    torch.cuda.synchronize()
    d = {}
    for i, t in cap["tokens"].items():
        d[f"tokens{i}"] = t
    for k, v in out.items():
        if torch.is_tensor(v) and k != "images":
            d[k] = v.clone()
    d["pose"] = out["pose_enc"][-1].clone()
    return d


def poison(value):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = [torch.full((64 << 20,), value, dtype=torch.float32, device="cuda") for _ in range(24)]   # 6 GiB
    small = [torch.full((n,), value, dtype=torch.float32, device="cuda") for n in (1 << 8, 1 << 12, 1 << 16, 1 << 18) for _ in range(64)]
    del blocks, small
    torch.cuda.synchronize()


ref = run(a)
for name, val in (("NaN", float("nan")), ("zeros", 0.0), ("1e4", 1e4), ("NaN again", float("nan"))):
    model.aggregator._ws._bufs.clear()          # workspaces too are re-allocated out of the poisoned pool
    poison(val)
    got = run(a)
    bad = []
    for k in ref:
        if not torch.equal(ref[k], got[k]):
            nan = int(torch.isnan(got[k]).sum())
            bad.append((k, f"nan={nan}" if nan else f"maxrel={float((ref[k] - got[k]).abs().max() / ref[k].abs().max()):.2e}"))
    print(f"free memory poisoned with {name:10s}:", bad if bad else "identical", flush=True)
