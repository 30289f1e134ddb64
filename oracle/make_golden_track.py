#!/usr/bin/env python3
"""Golden fixtures of the track head (query_points path), produced by the REFERENCE modules (build container only).

TEST INFRASTRUCTURE.  Usage:  python oracle/make_golden_track.py [--out DIR] [case ...]

Runs the reference aggregator and `TrackHead` (reference iggt/heads/track_head.py:75-109 as IGGT.forward calls it,
iggt/models/vggt.py:220-227) on seeded synthetic weights (oracle/weights.py with include_track=True) and seeded query
points, and stores: the tracker's feature maps, the coordinate predictions of all four refinement
iterations, visibility and confidence, plus two intermediates that pin the kernels one by one -- the sampled
correlation pyramid of the first iteration (input of `corr_mlp`) and the update transformer's first output.

The correlation pyramid has 7 levels of 2x average pooling, so the feature map (H/2 x W/2) must be at least 64 pixels
on a side: cases start at 140 x 140 (the reference raises in avg_pool2d below that).
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_shim, weights  # noqa: E402
from oracle.make_golden import GOLDEN_DIR, schema_of  # noqa: E402

# name: (S, H, W, weight mode, weight seed, image seed, number of query points)
CASES = {
    "track_s3_140_stress": (3, 140, 140, "stress", 0, 11, 24),
    "track_s2_140x182_stress": (2, 140, 182, "stress", 0, 12, 70),
}


def make_query_points(n, H, W, seed):
    """n seeded query points (x, y) in pixel coordinates of the H x W image, a few of them on / beyond the border."""
    u = weights.hash_uniform(2 * n, weights._name_seed("query_points", seed)).view(n, 2)
    q = torch.stack([u[:, 0] * (W - 1), u[:, 1] * (H - 1)], -1)
    if n >= 4:
        q[0] = torch.tensor([0.0, 0.0])
        q[1] = torch.tensor([W - 1.0, H - 1.0])
        q[2] = torch.tensor([W + 3.0, 0.5 * H])      # outside: border-clamped feature sample, zero-padded correlation
    return q


def run_case(model, name, out_dir):
    S, H, W, mode, wseed, iseed, nq = CASES[name]
    schema = schema_of(model)
    sd = weights.fill_state_dict(schema, seed=wseed, mode=mode, include_track=True)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(not schema[m]["dtype"].startswith("torch.float") for m in missing), missing
    images = weights.make_images(S, H, W, seed=iseed)[None]
    qp = make_query_points(nq, H, W, iseed)[None]            # [1, N, 2]
    cap = {}
    tr = model.track_head.tracker

    def keep_first(key, t):   # hooks must return None (anything else replaces the module's input / output)
        if key not in cap:
            cap[key] = t.detach().clone()

    hooks = [
        model.track_head.feature_extractor.register_forward_hook(lambda m, i, o: keep_first("fmaps", o)),
        tr.corr_mlp.register_forward_pre_hook(lambda m, i: keep_first("fcorrs_it0", i[0])),
        tr.updateformer.register_forward_hook(lambda m, i, o: keep_first("delta_it0", o[0])),
    ]
    t0 = time.time()
    with torch.no_grad():
        tokens, psi = model.aggregator(images)
        coords, vis, conf = model.track_head(tokens, images=images, patch_start_idx=psi, query_points=qp)
    for h in hooks:
        h.remove()
    print(f"[{name}] reference aggregator + track head {time.time() - t0:.1f}s", flush=True)
    out = {"meta": dict(S=S, H=H, W=W, mode=mode, weight_seed=wseed, image_seed=iseed, n_query=nq,
                        torch=torch.__version__),
           "query_points": qp.clone(),
           # complete, not sampled: the tracker-only parity test feeds the HIP tracker with the reference's own feature
           # maps (its flow embedding turns a 1e-4 pixel change of a coordinate into a 0.1 rad phase change, so the
           # fp32 tracker is pinned separately from the 16-bit-operand trunk in front of it)
           "fmaps": cap["fmaps"].clone(),                                      # [1,S,128,H/2,W/2]
           "fmaps_stats": dict(mean=float(cap["fmaps"].double().mean()), std=float(cap["fmaps"].double().std())),
           "fcorrs_it0": cap["fcorrs_it0"].clone(),                            # [N, S, 567]
           "delta_it0": cap["delta_it0"].clone(),                              # [1, N, S, 130]
           "coord_preds": torch.stack(coords, 0).clone(),                      # [4, 1, S, N, 2]
           "vis": vis.clone(), "conf": conf.clone()}                           # [1, S, N]
    path = os.path.join(out_dir, name + ".pt")
    torch.save(out, path)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)


def main():
    assert ref_shim.available(), "reference tree not found"
    argv = sys.argv[1:]
    out_dir = GOLDEN_DIR
    if argv[:1] == ["--out"]:
        out_dir = os.path.abspath(argv[1])
        argv = argv[2:]
    os.makedirs(out_dir, exist_ok=True)
    torch.manual_seed(0)
    model = ref_shim.build_reference_iggt(fast_init=True)
    ref_shim.assert_reference(type(model.track_head))
    ref_shim.assert_reference(type(model.track_head.tracker))
    for n in (argv or list(CASES)):
        run_case(model, n, out_dir)


if __name__ == "__main__":
    main()
