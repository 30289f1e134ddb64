"""Per-layer operand-precision table of the DPT heads' convolutions -- a SIMULATION, not product code.

The shipping conv kernels run every convolution as three bf16 MFMA passes (x_hi*w_hi + x_hi*w_lo + x_lo*w_hi, fp32
accumulate: csrc/conv_igemm.hip PREC = 3).  Cheaper operand formats would cost two passes or one.  What each of them would do
to the OUTPUT of the head is measured here without writing a kernel: the CPU restatement of the head (oracle/restate.py
`dpt_head`, pure torch) runs on the GPU in fp32 on the tokens the HIP trunk produced for a photograph fixture, with
`F.conv2d` replaced by a wrapper that rounds the operands of selected layers the way the candidate kernel would and sums
fp32 convolutions of the rounded parts (accumulation error is not modelled: fp32 accumulate is ~1e-6, far below every
operand effect in the table).

Variants (terms = MFMA passes at the fp16/bf16 rate):
  bf16x3   3  x_hi*w_hi + x_hi*w_lo + x_lo*w_hi, bf16 parts                       (ships)
  f16a_w2  2  fp16(x) * (w_hi + w_lo), fp16 parts of w                            activations rounded once
  a2_f16w  2  (x_hi + x_lo) * fp16(w) + mean-input compensation of the w rounding weights rounded once
  a2_nomc  2  (x_hi + x_lo) * fp16(w), no compensation
  f16x1mc  1  fp16(x) * fp16(w) + mean-input compensation (the trunk's scheme)
  f16x1    1  fp16(x) * fp16(w)
  bf16x1   1  bf16(x) * bf16(w)                                                   (PREC = 1 of the kernel)

Rows: (a) every layer alone (all others exact), sorted by its share of the head's conv FLOPs; (b) the variant on ALL layers;
(c) the variant on the layers that hold the top 50 % / 80 % of the FLOPs.  Columns: relative l2 error of the head output
(depth or points), fp32 head as the reference.  The last block runs the real HIP head on the same tokens.

    python probes/conv_precision.py [--case real_demo7_s4_crop518_stress] [--out gpurun_out/conv_precision.txt]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_REAL_CONV2D = F.conv2d


def _hi_lo(t, dt):
    hi = t.to(dt).float()
    return hi, (t - hi).to(dt).float()


def _mean_comp(x, w, w_r, stride, padding):
    """What the mean-input compensation adds: conv of the per-channel MEAN of x with the weight rounding residual (a
    per-output-channel constant away from the border; the border is handled exactly here, which flatters the variant a bit)."""
    mean = x.mean(dim=(0, 2, 3), keepdim=True).expand_as(x)
    return _REAL_CONV2D(mean, w - w_r, None, stride, padding)


def conv_variant(x, w, stride, padding, variant):
    c = lambda a, b: _REAL_CONV2D(a, b, None, stride, padding)
    if variant == "bf16x3":
        xh, xl = _hi_lo(x, torch.bfloat16)
        wh, wl = _hi_lo(w, torch.bfloat16)
        return c(xl, wh) + c(xh, wl) + c(xh, wh)
    if variant == "f16a_w2":
        xr = x.half().float()
        wh, wl = _hi_lo(w, torch.float16)
        return c(xr, wl) + c(xr, wh)
    if variant == "a2_f16w":
        xh, xl = _hi_lo(x, torch.float16)
        wr = w.half().float()
        return c(xl, wr) + c(xh, wr) + _mean_comp(x, w, wr, stride, padding)
    if variant == "a2_nomc":
        xh, xl = _hi_lo(x, torch.float16)
        wr = w.half().float()
        return c(xl, wr) + c(xh, wr)
    if variant == "f16x1mc":
        wr = w.half().float()
        return c(x.half().float(), wr) + _mean_comp(x, w, wr, stride, padding)
    if variant == "f16x1":
        return c(x.half().float(), w.half().float())
    if variant == "bf16x1":
        return c(x.bfloat16().float(), w.bfloat16().float())
    raise ValueError(variant)


VARIANTS = ["bf16x3", "f16a_w2", "a2_f16w", "a2_nomc", "f16x1mc", "f16x1", "bf16x1"]


class Hook:
    """F.conv2d stand-in: layers named in `self.active` run `self.variant`, everything else the real fp32 convolution."""

    def __init__(self, names):
        self.names = names          # id(weight) -> layer name
        self.active = set()
        self.variant = None
        self.flops = {}

    def __call__(self, x, w, bias=None, stride=1, padding=0, *a, **k):
        name = self.names.get(id(w), "?")
        if name in self.active:
            y = conv_variant(x, w, stride, padding, self.variant)
            if bias is not None:
                y = y + bias.view(1, -1, 1, 1)
        else:
            y = _REAL_CONV2D(x, w, bias, stride, padding, *a, **k)
        self.flops[name] = 2.0 * y.numel() * w.shape[1] * w.shape[2] * w.shape[3]
        return y


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def table(lines, sd, toks, H, W, S, hook, tail):
    from oracle import restate

    def run(head, act):
        with torch.no_grad():
            out, conf, _ = restate.dpt_head(sd, head, toks, H, W, act)
        return out, conf

    for head, act, key in (("depth_head", "exp", "depth"), ("point_head", "inv_log", "world_points")):
        hook.active = set()
        ref, ref_conf = run(head, act)
        flops = {k: v for k, v in hook.flops.items() if k.startswith(head)}
        hook.flops = {}
        total = sum(flops.values())
        order = sorted(flops, key=lambda k: -flops[k])
        lines += ["", f"## {head} ({key}); {len(order)} convolutions, {total * 1e-12:.3f} TFLOP for {S} views "
                      f"(transposed convolutions not included)"]
        lines.append(f"{'layer':46s} {'FLOP %':>7s} " + " ".join(f"{v:>9s}" for v in VARIANTS))
        for name in order:
            hook.active = {name}
            row = []
            for v in VARIANTS:
                hook.variant = v
                row.append(rel_l2(run(head, act)[0], ref))
            lines.append(f"{name[len(head) + 1:]:46s} {100 * flops[name] / total:7.2f} " + " ".join(f"{e:9.2e}" for e in row))
        sets = {"ALL layers": set(order)}
        for frac in (0.5, 0.8):
            acc, chosen = 0.0, set()
            for name in order:
                if acc >= frac * total:
                    break
                chosen.add(name)
                acc += flops[name]
            sets[f"top {int(frac * 100)} % of FLOPs ({len(chosen)} layers)"] = chosen
        sets["all BUT the top 50 %"] = set(order) - sets[[k for k in sets if k.startswith("top 50")][0]]
        for label, chosen in sets.items():
            hook.active = chosen
            row, row_conf, row_e2e = [], [], []
            for v in VARIANTS:
                hook.variant = v
                o, c = run(head, act)
                row.append(rel_l2(o, ref))
                row_conf.append(rel_l2(c, ref_conf))
                row_e2e.append(tail(key, None, o, None, e2e_only=True))
            lines.append(f"{label:46s} {100 * sum(flops[n] for n in chosen) / total:7.2f} " + " ".join(f"{e:9.2e}" for e in row))
            lines.append(f"{'   ... confidence map':46s} {'':7s} " + " ".join(f"{e:9.2e}" for e in row_conf))
            if row_e2e[0] is not None:
                lines.append(f"{'   ... trunk + this head vs reference fixture':46s} {'':7s} " + " ".join(f"{e:9.2e}" for e in row_e2e))
        hook.active = set()
        tail(key, lines, ref, ref_conf)


def selftest():
    from helpers import schema
    from oracle import restate, weights

    torch.manual_seed(0)
    S, H, W = 2, 56, 56
    sd = weights.fill_state_dict(schema(), seed=0, mode="stress", device="cpu", include_track=False)
    with torch.no_grad():
        toks = restate.aggregator(sd, torch.rand(S, 3, H, W))
    sd = {k: v for k, v in sd.items() if k.startswith(("depth_head.", "point_head."))}
    hook = Hook({id(v): k[: -len(".weight")] for k, v in sd.items() if k.endswith(".weight") and v.dim() == 4})
    F.conv2d = hook
    lines = []
    table(lines, sd, toks, H, W, S, hook, lambda key, lines, ref, conf, e2e_only=False: None)
    F.conv2d = _REAL_CONV2D
    print("\n".join(lines))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="real_demo7_s4_crop518_stress")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "conv_precision.txt"))
    ap.add_argument("--cpu-selftest", action="store_true", help="plumbing check without a GPU: tiny images, tokens from the restated trunk")
    args = ap.parse_args()
    if args.cpu_selftest:
        return selftest()

    from conftest import load_golden
    from helpers import build_gpu_model, schema
    from iggt.utils.load_fn import load_and_preprocess_images
    from iggt_official_amd import precision
    from oracle import restate, weights

    g = load_golden(args.case)
    m = g["meta"]
    paths = [os.path.join(ROOT, "tests", "golden", "images", m["scene"], f) for f in m["files"]]
    tgt = m["resize_target_size"]
    images = load_and_preprocess_images(paths, mode=m["loader_mode"], resize_target_size=None if tgt is None else tuple(tgt))
    H, W = m["H"], m["W"]
    model = build_gpu_model(m["mode"], m["weight_seed"])
    precision.set_operand_dtype("f16")
    cap = {}
    h = model.aggregator.register_forward_hook(lambda mod, i, o: cap.__setitem__("tokens", o[0]))
    with torch.no_grad():
        pred = model(images)
    h.remove()
    toks = [None if t is None else t.float() for t in cap["tokens"]]
    sd = weights.fill_state_dict(schema(), seed=m["weight_seed"], mode=m["mode"], device="cuda", include_track=False)
    sd = {k: v for k, v in sd.items() if k.startswith(("depth_head.", "point_head."))}

    cpu_pos = restate.pos_embed
    restate.pos_embed = lambda *a, **k: cpu_pos(*a, **k).cuda()
    hook = Hook({id(v): k[: -len(".weight")] for k, v in sd.items() if k.endswith(".weight") and v.dim() == 4})
    F.conv2d = hook
    torch.backends.cudnn.benchmark = False

    lines = [f"# probes/conv_precision.py --case {args.case}: S = {m['S']} photographs {H} x {W}, stress weights; tokens from the HIP",
             "# trunk (fp16 operands); head = oracle/restate.py dpt_head in fp32 on the GPU with per-layer operand rounding SIMULATED.",
             "# Entries: relative l2 error of the head output against the all-fp32 head.  terms = MFMA passes the variant would cost.",
             "# terms: bf16x3 3 (ships) | f16a_w2 2 | a2_f16w 2 | a2_nomc 2 | f16x1mc 1 | f16x1 1 | bf16x1 1"]

    ss = m["spatial_stride"]

    def tail(key, lines, ref, ref_conf, e2e_only=False):
        if e2e_only:
            return rel_l2(ref[0][:, ::ss, ::ss].cpu(), g[key][0])
        hip, hip_conf = pred[key][0], pred[key + "_conf"][0]
        lines.append(f"{'HIP head as shipped (same tokens), measured':46s} {'':7s} {rel_l2(hip, ref[0]):9.2e}   conf {rel_l2(hip_conf, ref_conf[0]):9.2e}")
        lines.append(f"{'HIP end to end vs the reference fixture':46s} {'':7s} {rel_l2(hip[:, ::ss, ::ss].cpu(), g[key][0]):9.2e}"
                     f"   (trunk + head; the 1e-3 budget applies to this number)")
        lines.append(f"{'fp32 head on HIP tokens vs the reference fixture':46s} {'':7s} {rel_l2(ref[0][:, ::ss, ::ss].cpu(), g[key][0]):9.2e}"
                     f"   (what the trunk alone contributes)")

    table(lines, sd, toks, H, W, m["S"], hook, tail)
    F.conv2d = _REAL_CONV2D
    text = "\n".join(lines) + "\n"
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write(text)
    print(text)


if __name__ == "__main__":
    main()
