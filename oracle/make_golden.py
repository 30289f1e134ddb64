#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE modules (build container only).

TEST INFRASTRUCTURE.  Usage:  python oracle/make_golden.py [--out DIR] [case ...]
(default cases: every one except the BASELINE-size ones in LARGE, which take minutes to an hour of CPU time and are
generated on request: `python oracle/make_golden.py full_s8_518_stress full_s32_518_stress`)

Imports /root/reference through oracle/ref_shim.py, fills the reference IGGT with the
seeded synthetic weights of oracle/weights.py, runs the forward path exactly as
IGGT.forward orchestrates it (reference iggt/models/vggt.py:185-218; sub-modules are
called directly so that odd patch grids / S>12 can be covered)
and writes tests/golden/<case>.pt plus tests/golden/state_dict_schema.json.

Large cases store strided samples of the dense maps so the fixtures stay small (meta: spatial_stride for the
[H, W] maps, token_stride / channel_stride for the token tensors); tests index the HIP output with the same strides.
Whole-tensor statistics (mean / abs-sum / std of every dense output) pin the un-sampled part.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_shim, weights  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name: (S, H, W, weight mode, weight seed, image seed, spatial sample stride, token sample stride[, channel stride])
CASES = {
    "tiny_s2_56_stress": (2, 56, 56, "stress", 0, 1, 1, 1),
    "tiny_s3_84x56_stress": (3, 84, 56, "stress", 0, 2, 1, 1),
    "tiny_s2_56_default": (2, 56, 56, "default", 0, 1, 1, 1),
    "tiny_s2_70_stress": (2, 70, 70, "stress", 0, 3, 1, 1),  # odd grid: geometry outputs only
    "tiny_s5_112_stress": (5, 112, 112, "stress", 0, 4, 2, 1),
    "full_s2_518_stress": (2, 518, 518, "stress", 0, 5, 7, 16),
    "demo_s3_336x504_stress": (3, 336, 504, "stress", 0, 6, 7, 16),
    # BASELINE.json configs[1] / configs[2]: 8 and 32 views @ 518^2 (N_global = 10 992 / 43 968)
    "full_s8_518_stress": (8, 518, 518, "stress", 0, 7, 7, 32, 4),
    "full_s32_518_stress": (32, 518, 518, "stress", 0, 8, 11, 97, 4),
    # BASELINE.json configs[4]'s per-view shape: 1036^2 (74 x 74 patch grid, 5 481 tokens per view; part head valid)
    "full_s2_1036_stress": (2, 1036, 1036, "stress", 0, 9, 14, 64, 4),
}
LARGE = ("full_s8_518_stress", "full_s32_518_stress", "full_s2_1036_stress")
# The reference's part head evaluates `cross_attention_1` (whose result it discards, part_head.py:178-185) with an explicit
# softmax over (4g)^2 x (4g)^2 scores per frame and head: 87 616^2 x 8 x 4 B = 245 GB at 1036^2 -- it cannot run here.
NO_PART = ("full_s2_1036_stress",)


def schema_of(model):
    sd = model.state_dict()
    return {k: {"shape": list(v.shape), "dtype": str(v.dtype)} for k, v in sd.items()}


def run_case(model, name):
    S, H, W, mode, wseed, iseed, sstride, tstride = CASES[name][:8]
    cstride = CASES[name][8] if len(CASES[name]) > 8 else 1
    schema = schema_of(model)
    t0 = time.time()
    sd = weights.fill_state_dict(schema, seed=wseed, mode=mode)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m.startswith("track_head.") or not schema[m]["dtype"].startswith("torch.float")
               for m in missing), missing
    print(f"[{name}] weights filled in {time.time() - t0:.1f}s", flush=True)

    images = weights.make_images(S, H, W, seed=iseed)[None]  # [1,S,3,H,W]
    out = {"meta": dict(S=S, H=H, W=W, mode=mode, weight_seed=wseed, image_seed=iseed,
                        spatial_stride=sstride, token_stride=tstride, channel_stride=cstride,
                        torch=torch.__version__)}
    cap = {}
    h = model.aggregator.patch_embed.register_forward_hook(
        lambda m, i, o: cap.__setitem__("dino", o["x_norm_patchtokens"].detach().clone()))
    t0 = time.time()
    with torch.no_grad():
        tokens, psi = model.aggregator(images)
        h.remove()
        if S > 12:   # keep only what the heads read (layers 4/11/17/23): 24 x 360 MB at 32 views otherwise
            tokens = [t if i in (4, 11, 17, 23) else None for i, t in enumerate(tokens)]
        print(f"[{name}] reference aggregator {time.time() - t0:.1f}s", flush=True)
        pose = model.camera_head(tokens)
        depth, depth_conf = model.depth_head(tokens, images=images, patch_start_idx=psi, frames_chunk_size=None)
        pts, pts_conf, point_feat = model.point_head(tokens, images=images, patch_start_idx=psi,
                                                     frames_chunk_size=None)
        part_ok = (H % 28 == 0) and (W % 28 == 0) and name not in NO_PART
        if part_ok:
            ada, _pos = model.part_adaptor(tokens, images=images, patch_start_idx=psi)
            part = model.part_head(list(ada.values()), point_feature=point_feat, images=images,
                                   patch_start_idx=psi, frames_chunk_size=None)
    print(f"[{name}] reference forward {time.time() - t0:.1f}s", flush=True)

    def sp(t, dims):  # strided spatial sample over dims (h, w)
        if sstride == 1:
            return t.clone()
        idx = [slice(None)] * t.ndim
        for d in dims:
            idx[d] = slice(0, None, sstride)
        return t[tuple(idx)].clone()

    ts = tstride
    cs = cstride
    out["dino"] = cap["dino"][:, ::ts, ::cs].clone()                 # [S, g2/ts, 1024/cs]
    for li in (4, 11, 17, 23):
        out[f"tokens_{li}"] = tokens[li][:, :, ::ts, ::cs].clone()    # [1,S,P/ts,2048/cs]
    out["tokens_23_special"] = tokens[23][:, :, :5].clone()
    out["pose_enc"] = torch.stack(pose, 0)                            # [4,1,S,9]
    out["depth"] = sp(depth, (2, 3))
    out["depth_conf"] = sp(depth_conf, (2, 3))
    out["world_points"] = sp(pts, (2, 3))
    out["world_points_conf"] = sp(pts_conf, (2, 3))
    if sstride == 1:
        for i, f in enumerate(point_feat):
            out[f"point_feat_{i}"] = f.clone()
    if part_ok:
        out["part_feat"] = sp(part, (3, 4))
        if sstride == 1:
            for k, v in ada.items():
                out[f"adaptor_{k}"] = v.clone()
    # whole-tensor statistics pin the un-sampled part too
    stats = {}
    for k, v in [("depth", depth), ("depth_conf", depth_conf), ("world_points", pts),
                 ("world_points_conf", pts_conf)] + ([("part_feat", part)] if part_ok else []):
        stats[k] = dict(mean=float(v.double().mean()), abs_sum=float(v.double().abs().sum()),
                        std=float(v.double().std()))
    out["stats"] = stats
    path = os.path.join(OUT_DIR, name + ".pt")
    torch.save(out, path)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)


OUT_DIR = GOLDEN_DIR


def main():
    global OUT_DIR
    assert ref_shim.available(), "reference tree not found"
    argv = sys.argv[1:]
    if argv[:1] == ["--out"]:
        OUT_DIR = os.path.abspath(argv[1])
        argv = argv[2:]
    os.makedirs(OUT_DIR, exist_ok=True)
    torch.manual_seed(0)
    t0 = time.time()
    model = ref_shim.build_reference_iggt(fast_init=True)
    print(f"reference IGGT built in {time.time() - t0:.1f}s", flush=True)
    schema = schema_of(model)
    with open(os.path.join(OUT_DIR, "state_dict_schema.json"), "w") as f:
        json.dump(schema, f, indent=0, sort_keys=True)
    # integer buffers (relative position indices) are structural, save their values
    ints = {k: v.clone() for k, v in model.state_dict().items() if not v.dtype.is_floating_point and v.numel() > 1}
    torch.save(ints, os.path.join(OUT_DIR, "int_buffers.pt"))
    names = argv or [c for c in CASES if c not in LARGE]
    for n in names:
        run_case(model, n)


if __name__ == "__main__":
    main()
