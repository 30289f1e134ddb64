"""Which GROUPS of fp16 rounding sites carry the token error at a given weight dose?  (CPU, analysis tool; round 5, review item 1)

Leave-one-OUT ablation (oracle/precision_sim.py --ablate) cannot answer that on the heavy-tailed doses: no single site dominates,
removing one of ~12 comparable contributions changes the total by a few per cent.  This probe runs leave-one-IN: only the named
group of sites is rounded to fp16, everything else stays fp32 -- the contributions then add in quadrature (last column checks it).

    python probes/precision_groups.py [S H W] mode [mode ...]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import precision_sim, weights  # noqa: E402

ALL = precision_sim.SITES + ("w_qkv", "w_proj", "w_fc1", "w_fc2")
GROUPS = {
    "weights": ("weights",),                 # (the four w_* names only matter while "weights" is rounded)
    "ln_out (xn1,xn2)": ("xn1", "xn2"),
    "qkv_out": ("qkv",),
    "qk_post_norm": ("qk",),
    "p": ("p",),
    "attn_out (o)": ("o",),
    "hid": ("hid",),
}


def main():
    args = sys.argv[1:]
    dims = [int(a) for a in args[:3]] if len(args) >= 3 and args[0].isdigit() else None
    modes = args[3:] if dims else args
    S, H, W = dims or (2, 56, 56)
    schema = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "state_dict_schema.json")))
    images = weights.make_images(S, H, W, seed=2, device="cpu")
    for mode in modes:
        sd = weights.fill_state_dict(schema, seed=1, mode=mode, device="cpu")
        sd = {k: v for k, v in sd.items() if k.startswith("aggregator.")}
        with torch.no_grad():
            ref = precision_sim.run(sd, images, None)[23]
            full = precision_sim.run(sd, images, "fp16")[23]
            e_full = float((full - ref).norm() / ref.norm())
            print(f"# {mode}: {S} views @ {H}x{W}; all sites fp16: {e_full:.2e}", flush=True)
            ss = 0.0
            for name, sites in GROUPS.items():
                exact = tuple(s for s in ALL if s not in sites and not (s.startswith("w_") and "weights" in sites))
                out = precision_sim.run(sd, images, "fp16", exact=exact)[23]
                e = float((out - ref).norm() / ref.norm())
                ss += e * e
                print(f"   only {name:20s} rounded: {e:.2e}", flush=True)
            print(f"   quadrature sum of the groups: {ss ** 0.5:.2e}", flush=True)


if __name__ == "__main__":
    main()
