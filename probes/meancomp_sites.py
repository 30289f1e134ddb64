#!/usr/bin/env python3
"""Which of a block's four GEMMs need the mean-input compensation?  Token-layer error on a photograph fixture (demo7, 4 views
@ 518^2) and on the 8-view hash-noise fixture for every subset of {qkv, proj, fc1, fc2}."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from conftest import GOLDEN  # noqa: E402
from helpers import build_gpu_model, errors  # noqa: E402
from iggt.utils.load_fn import load_and_preprocess_images  # noqa: E402
from iggt_official_amd import precision  # noqa: E402
from oracle import weights  # noqa: E402

model = build_gpu_model("stress", 0)
cases = []
g = torch.load(os.path.join(GOLDEN, "real_demo7_s4_crop518_stress.pt"), weights_only=False)
m = g["meta"]
cases.append(("photo demo7 4x518", g, load_and_preprocess_images([os.path.join(GOLDEN, "images", m["scene"], f) for f in m["files"]],
                                                                  mode=m["loader_mode"])))
g2 = torch.load(os.path.join(GOLDEN, "full_s8_518_stress.pt"), weights_only=False)
m2 = g2["meta"]
cases.append(("noise 8x518", g2, weights.make_images(m2["S"], m2["H"], m2["W"], seed=m2["image_seed"], device="cuda")))
sites = ("qkv", "proj", "fc1", "fc2")
print(f"{'sites':22s} " + " ".join(f"{c[0]:>38s}" for c in cases))
for r in range(5):
    for sub in itertools.combinations(sites, r):
        precision.set_mean_compensation_sites(sub)
        row = []
        for name, gg, img in cases:
            mm = gg["meta"]
            cap = {}
            h = model.aggregator.register_forward_hook(lambda mod, i, o: cap.__setitem__("t", o[0]))
            pred = model(img)
            h.remove()
            torch.cuda.synchronize()
            ts, cs, ss = mm["token_stride"], mm.get("channel_stride", 1), mm["spatial_stride"]
            tok = max(errors(cap["t"][li][:, :, ::ts, ::cs], gg[f"tokens_{li}"])[1] for li in (4, 11, 17, 23))
            dep = errors(pred["depth"][:, :, ::ss, ::ss], gg["depth"])
            wp = errors(pred["world_points"][:, :, ::ss, ::ss], gg["world_points"])
            row.append(f"tok {tok:.2e} depth {dep[1]:.2e}/{dep[2]:.2e} pts {wp[1]:.2e}")
        print(f"{','.join(sub) or '-':22s} " + " ".join(f"{x:>38s}" for x in row), flush=True)
