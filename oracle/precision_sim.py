"""What operand format would the trunk need for the 1e-3 parity target?  (analysis tool -- test infrastructure)

Re-runs the CPU restatement of the trunk (DINOv2 + 24 x (frame, global) blocks) with the SAME rounding points as
the HIP path -- GEMM weights, LayerNorm output, q/k/v, the softmax numerators P, the attention output and the MLP
hidden activation are rounded to a 16-bit format, everything else (accumulation, LayerNorm, RoPE, softmax
statistics, residual stream) stays fp32 -- and reports the relative l2 error of the aggregated tokens against
plain fp32 for bf16 (8 significant bits) and fp16 (11 bits).

    python -m oracle.precision_sim [S] [H] [W] [--ablate] [--mode=stress|default|trained_like ...]
    (--ablate: leave one rounding site at a time in fp32)
"""
import sys

import torch
import torch.nn.functional as F

from . import restate, weights

FMT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def _round(t, fmt):
    """fmt "fp16x2" (round 5): the value as an fp16 pair hi + lo, what the escalated blocks' three-pass MFMA products see
    (layers/blocks.py `x3`); subnormal lo parts lose bits exactly as on the device."""
    if fmt in ("fp16x2", "fp16x2-p1"):
        hi = t.to(torch.float16).float()
        return hi + (t - hi).to(torch.float16).float()
    return t.to(FMT[fmt]).float()


SITES = ("weights", "xn1", "qkv", "qk", "p", "o", "xn2", "hid")


def run(sd, images, fmt, exact=(), per_block=None):
    """exact: rounding sites left in fp32 (ablation).  per_block (round 5): callable(block prefix, e.g.
    "aggregator.global_blocks.3") -> format name for THAT block (None: `fmt`) -- the per-block precision rung."""
    def sites_for(f):
        def site(name):
            if f is None or name in exact:
                return lambda t: t
            if f == "fp16x2-p1" and name == "p":    # variant: the softmax numerators stay single fp16
                return lambda t: t.to(torch.float16).float()
            return lambda t: _round(t, f)
        return {n: site(n) for n in SITES[1:]}

    cache = {}

    def fmt_of(prefix):
        f = per_block(prefix) if per_block is not None else None
        return fmt if f is None else f

    def R(prefix):
        f = fmt_of(prefix)
        if f not in cache:
            cache[f] = sites_for(f)
        return cache[f]

    sdw = dict(sd)
    if "weights" not in exact:
        kinds = tuple(f".{n}.weight" for n in ("qkv", "proj", "fc1", "fc2") if "w_" + n not in exact)
        for k, v in sd.items():
            if kinds and k.endswith(kinds) and v.dim() >= 2:
                # (the DINOv2 patch-embed convolution ends in ".proj.weight" too: it follows the format of "<vit>.patch_embed")
                blk = k.split(".attn.")[0].split(".mlp.")[0] if (".attn." in k or ".mlp." in k) else k.rsplit(".proj.weight", 1)[0]
                f = fmt_of(blk)
                if f is not None:
                    sdw[k] = _round(v, f)

    def attention(sd_, p, x, heads, pos=None, r=None):
        B, N, C = x.shape
        qkv = r["qkv"](restate._lin(sd_, p + ".qkv", x)).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        if p + ".q_norm.weight" in sd_:
            q, k = restate._ln(sd_, p + ".q_norm", q), restate._ln(sd_, p + ".k_norm", k)
        if pos is not None:
            q, k = restate.rope2d(q, pos), restate.rope2d(k, pos)
        q, k = r["qk"](q), r["qk"](k)
        s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
        pnum = torch.exp(s - s.amax(-1, keepdim=True))
        o = r["o"]((r["p"](pnum) @ v) / pnum.sum(-1, keepdim=True))
        return restate._lin(sd_, p + ".proj", o.transpose(1, 2).reshape(B, N, C))

    def block(sd_, p, x, heads, pos=None, eps=1e-5):
        r = R(p)
        x = x + sd_[p + ".ls1.gamma"] * attention(sd_, p + ".attn", r["xn1"](restate._ln(sd_, p + ".norm1", x, eps)), heads, pos, r)
        h = r["hid"](F.gelu(restate._lin(sd_, p + ".mlp.fc1", r["xn2"](restate._ln(sd_, p + ".norm2", x, eps)))))
        return x + sd_[p + ".ls2.gamma"] * restate._lin(sd_, p + ".mlp.fc2", h)

    old = restate.block
    restate.block = block
    try:
        return restate.aggregator(sdw, images)
    finally:
        restate.block = old


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    S, H, W = (int(a) for a in (args[:3] + ["2", "56", "56"][len(args):]))
    import json, os
    schema = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "state_dict_schema.json")))
    modes = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--mode=")] or ["stress", "default"]
    for mode in modes:
        sd = weights.fill_state_dict(schema, seed=1, mode=mode, device="cpu")
        sd = {k: v for k, v in sd.items() if k.startswith("aggregator.")}
        images = weights.make_images(S, H, W, seed=2, device="cpu")
        with torch.no_grad():
            ref = run(sd, images, None)
            for fmt in ("bf16", "fp16"):
                out = run(sd, images, fmt)
                errs = {i: float((out[i] - ref[i]).norm() / ref[i].norm()) for i in ref}
                print(f"weights={mode:8s} operands={fmt}: token l2 error per kept layer "
                      + " ".join(f"{i}:{e:.2e}" for i, e in errs.items()), flush=True)
            if mode != "default" and "--ablate" in sys.argv:
                for name in SITES + ("w_qkv", "w_proj", "w_fc1", "w_fc2"):
                    out = run(sd, images, "fp16", exact=(name,))
                    e = float((out[23] - ref[23]).norm() / ref[23].norm())
                    print(f"   fp16 with site '{name}' exact: layer-23 token error {e:.2e}", flush=True)


if __name__ == "__main__":
    main()
