"""Which ingredient of the `trained_like` synthetic checkpoint makes the trunk ill-conditioned?  (CPU, analysis tool)

For each weight mode: the CPU restatement of the aggregator in fp64, in fp32 and with the HIP path's fp16 rounding sites
(oracle/precision_sim.py); reports the relative l2 distance of the layer-23 tokens
   fp32 vs fp64      -> how far the REFERENCE's own fp32 arithmetic sits from the exact map (conditioning of the weights)
   fp16-sim vs fp32  -> what the 16-bit operand path can reach at best (no mean compensation in the simulation)
and the spread of the global-attention logits (std / max over a block) that the q/k-norm affines produce.

    python probes/trained_like_sweep.py [S H W] mode [mode ...]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import precision_sim, restate, weights  # noqa: E402


def main():
    args = sys.argv[1:]
    dims = [int(a) for a in args[:3]] if len(args) >= 3 and args[0].isdigit() else None
    modes = args[3:] if dims else args
    S, H, W = dims or (2, 56, 56)
    schema = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "state_dict_schema.json")))
    images = weights.make_images(S, H, W, seed=2, device="cpu")
    print(f"# {S} views @ {H}x{W}; layer-23 token l2 distances")
    for mode in modes:
        sd = weights.fill_state_dict(schema, seed=1, mode=mode, device="cpu")
        sd = {k: v for k, v in sd.items() if k.startswith("aggregator.")}
        logit = {}
        old_attn = restate.attention

        def spy(sd_, p, x, heads, pos=None):
            if ".global_blocks.12." in p and "s" not in logit:
                B, N, C = x.shape
                qkv = restate._lin(sd_, p + ".qkv", x).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
                q, k = restate._ln(sd_, p + ".q_norm", qkv[0]), restate._ln(sd_, p + ".k_norm", qkv[1])
                s = (q @ k.transpose(-1, -2)) * 0.125
                pr = torch.softmax(s.float(), -1)
                logit["s"] = (float(s.std()), float(s.amax()), float(pr.amax(-1).mean()))
            return old_attn(sd_, p, x, heads, pos)

        with torch.no_grad():
            restate.attention = spy
            try:
                r32 = restate.aggregator(sd, images)[23]
            finally:
                restate.attention = old_attn
            sd64 = {k: v.double() for k, v in sd.items()}
            r64 = restate.aggregator(sd64, images.double())[23]
            r16 = precision_sim.run(sd, images, "fp16")[23]
            rb16 = precision_sim.run(sd, images, "bf16")[23]
        e3264 = float((r32.double() - r64).norm() / r64.norm())
        e1632 = float((r16 - r32).norm() / r32.norm())
        eb1632 = float((rb16 - r32).norm() / r32.norm())
        ls = logit.get("s", (0, 0, 0))
        print(f"{mode:60s} fp32|fp64 {e3264:.2e}   fp16sim|fp32 {e1632:.2e}   bf16sim|fp32 {eb1632:.2e}   logits(global 12): std {ls[0]:.1f} max {ls[1]:.0f} "
              f"mean top-prob {ls[2]:.2f}", flush=True)


if __name__ == "__main__":
    main()
