"""hipGraph replay of the forward (iggt_official_amd/graphs.py): captured once per input shape, bit-identical to the
eager forward, static outputs, dropped when parameters are reloaded."""
import pytest
import torch

from conftest import load_golden
from helpers import build_gpu_model, errors

pytestmark = pytest.mark.gpu


def test_graphed_forward_equals_eager_and_tracks_new_inputs():
    from oracle import weights

    g = load_golden("tiny_s2_56_stress")
    m = g["meta"]
    model = build_gpu_model(m["mode"], m["weight_seed"])
    a = weights.make_images(m["S"], m["H"], m["W"], seed=m["image_seed"], device="cuda")
    b = weights.make_images(m["S"], m["H"], m["W"], seed=31, device="cuda")
    eager_a = {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in model(a).items()}
    eager_b = {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in model(b).items()}
    try:
        model.enable_graphs(True)
        for img, ref in ((a, eager_a), (b, eager_b), (a, eager_a)):
            out = model(img)
            torch.cuda.synchronize()
            for k in ("depth", "depth_conf", "world_points", "world_points_conf", "part_feat"):
                assert torch.equal(out[k], ref[k]), k
            assert all(torch.equal(x, y) for x, y in zip(out["pose_enc"], ref["pose_enc"]))
        assert len(model._gcache._graphs) == 1
        seg = next(iter(model._gcache._graphs.values()))[1]
        assert seg.num_segments == 1                     # single GPU: no collective, one graph
        assert errors(out["depth"], g["depth"])[1] < 1e-3
        # another shape -> another graph; reloading parameters drops all of them
        c = weights.make_images(3, 56, 84, seed=5, device="cuda")
        out_c = model(c)
        assert out_c["depth"].shape == (1, 3, 56, 84, 1) and len(model._gcache._graphs) == 2
        model.load_state_dict(model.state_dict(), strict=False)
        assert len(model._gcache._graphs) == 0
    finally:
        model.enable_graphs(False)
