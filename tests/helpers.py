"""Shared test helpers (GPU model construction with the seeded synthetic weights)."""
import json
import os

import torch

from conftest import GOLDEN

_MODELS = {}


def schema():
    with open(os.path.join(GOLDEN, "state_dict_schema.json")) as f:
        return json.load(f)


def build_gpu_model(mode="stress", seed=0, part_on_invalid_grid="skip", include_track=False):
    """IGGT on cuda:0 filled with oracle.weights synthetic parameters (generated on-device by the
    integer hash, bit-identical to the CPU values used for the golden fixtures).  The track head keeps its
    constructor initialisation unless include_track (it only runs when query_points are passed)."""
    key = (mode, seed, part_on_invalid_grid, include_track)
    if key in _MODELS:
        from iggt_official_amd import precision

        precision.reset_guards(_MODELS[key])    # outputs must not depend on what an earlier test fed this model
        return _MODELS[key]
    from iggt.models.vggt import IGGT
    from oracle import weights

    _MODELS.clear()  # one 1.3 B-parameter model resident at a time
    with torch.device("cuda"):
        model = IGGT(part_on_invalid_grid=part_on_invalid_grid).eval()
    sd = weights.fill_state_dict(schema(), seed=seed, mode=mode, device="cuda", include_track=include_track)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("relative_position_index" in m or "num_batches_tracked" in m
               or (m.startswith("track_head.") and not include_track) for m in missing), missing
    _MODELS[key] = model
    return model


def errors(got, ref):
    """(max|d| / max|ref|, ||d|| / ||ref||, ||d|| / ||ref - mean(ref)||) in float64."""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    d = got - ref
    return (float(d.abs().max() / ref.abs().max().clamp_min(1e-30)),
            float(d.norm() / ref.norm().clamp_min(1e-30)),
            float(d.norm() / (ref - ref.mean()).norm().clamp_min(1e-30)))


class FeatureTap:
    """Captures the part branch's INPUTS during a forward: the SamProjector pyramid (part_adaptor -> res1..res4, NCHW-shaped
    views) and the point head's fusion features (out2, out3, out4; NHWC maps) -- reference vggt.py:204-218 hands exactly
    these to the part head.  `compare(g, meta)` returns errors() per fixture key (`adaptor_res*`, `point_feat_*`), sampled
    like oracle/make_golden.py stored them (meta["feature_sample"] = (channel stride, row / column stride); whole maps in the
    small fixtures)."""

    def __init__(self, model):
        self.cap = {}
        self._h = [model.part_adaptor.register_forward_hook(lambda m, i, o: self.cap.__setitem__("ada", o[0])),
                   model.point_head.register_forward_hook(lambda m, i, o: self.cap.__setitem__("pf", o[2]))]

    def remove(self):
        for h in self._h:
            h.remove()

    def compare(self, g, meta):
        fsamp = meta.get("feature_sample")
        cs, fs = (1, 1) if fsamp is None else (int(fsamp[0]), int(fsamp[1]))
        res = {}
        for k, v in self.cap.get("ada", {}).items():
            if f"adaptor_{k}" in g:
                res[f"adaptor_{k}"] = errors(v[:, ::cs, ::fs, ::fs], g[f"adaptor_{k}"])
        for i, f in enumerate(self.cap.get("pf", ())):
            if f"point_feat_{i}" in g:
                res[f"point_feat_{i}"] = errors(f.permute(0, 3, 1, 2)[:, ::cs, ::fs, ::fs], g[f"point_feat_{i}"])
        return res
