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


def _copy(pred):
    return {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in pred.items()}


def _same(out, ref):
    for k in ("depth", "depth_conf", "world_points", "world_points_conf", "part_feat"):
        assert torch.equal(out[k], ref[k]), k
    assert all(torch.equal(x, y) for x, y in zip(out["pose_enc"], ref["pose_enc"]))


def test_graph_survives_workspace_growth_and_operand_round_trip():
    """A captured graph holds raw pointers into the block engine's workspaces and weight packs.  Capturing a LARGER shape grows
    the workspaces (the old buffers are freed), switching the operand format re-packs every weight: the small graph must then
    be re-captured instead of replaying into freed memory (graphs.py allocation generation).  Sequence A, B (larger), A, then
    f16 -> bf16 -> f16, every replay bit-identical to the eager forward."""
    from iggt_official_amd import graphs, precision
    from oracle import weights

    model = build_gpu_model("stress", 0)
    a = weights.make_images(2, 56, 56, seed=1, device="cuda")
    b = weights.make_images(3, 112, 84, seed=5, device="cuda")
    precision.set_operand_dtype("f16")
    model.aggregator._ws._bufs.clear()                # start from small workspaces whatever ran before
    eager_a, eager_b = _copy(model(a)), None
    model.aggregator._ws._bufs.clear()
    try:
        model.enable_graphs(True)
        _same(model(a), eager_a)
        n0 = model._gcache.captures
        gen = graphs.alloc_generation()
        out_b = _copy(model(b))                       # larger shape: the workspaces are reallocated during its warm-up
        assert graphs.alloc_generation() > gen and model._gcache.captures == n0 + 1
        # junk into the memory the allocator got back, so that a stale replay would be visible
        junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(64)]
        _same(model(a), eager_a)                      # stale entry detected -> re-captured
        assert model._gcache.captures == n0 + 2
        _same(model(a), eager_a)                      # and now a plain replay
        assert model._gcache.captures == n0 + 2
        _same(model(b), out_b)                        # B was captured after the growth: still valid, no new capture
        assert model._gcache.captures == n0 + 2
        del junk
        model.enable_graphs(False)
        eager_b = _copy(model(b))
        _same(out_b, eager_b)
        model.enable_graphs(True)
        _same(model(a), eager_a)
        n1 = model._gcache.captures
        precision.set_operand_dtype("bf16")
        bf = _copy(model(a))
        assert not torch.equal(bf["depth"], eager_a["depth"]) and errors(bf["depth"], eager_a["depth"])[1] < 2e-2
        precision.set_operand_dtype("f16")
        _same(model(a), eager_a)                      # the f16 graph of before the round trip points at freed packs: re-captured
        assert model._gcache.captures >= n1 + 2
    finally:
        precision.set_operand_dtype("f16")
        model.enable_graphs(False)


def test_batched_scenes_with_graphs():
    """B = 2 scenes under enable_graphs(): both scenes replay the same graph, whose outputs are static buffers -- each scene's
    outputs are cloned before the next replay (models/vggt.py _scenes)."""
    from oracle import weights

    model = build_gpu_model("stress", 0)
    x0 = weights.make_images(2, 56, 56, seed=1, device="cuda")
    x1 = weights.make_images(2, 56, 56, seed=77, device="cuda")
    e0, e1 = _copy(model(x0)), _copy(model(x1))
    try:
        model.enable_graphs(True)
        both = model(torch.stack([x0, x1], 0))
        for k in ("depth", "world_points", "part_feat"):
            assert torch.equal(both[k][0], e0[k][0]) and torch.equal(both[k][1], e1[k][0]), k
        assert torch.equal(both["pose_enc"][-1][0], e0["pose_enc"][-1][0])
        assert torch.equal(both["pose_enc"][-1][1], e1["pose_enc"][-1][0])
    finally:
        model.enable_graphs(False)


def test_depth_head_on_a_side_stream_changes_nothing():
    """models/vggt.py runs the depth head beside the point head on a second side stream for large inputs (IGGT_HEAD_STREAMS, size
    rule).  Forced on at a small shape: outputs bit-identical to the in-line order, eagerly (repeated: a race would not be) and as a
    captured graph (fork / join inside the capture)."""
    from iggt_official_amd.models import vggt as mv
    from oracle import weights

    g = load_golden("demo_s3_336x504_stress")
    m = g["meta"]
    model = build_gpu_model(m["mode"], m["weight_seed"])
    img = weights.make_images(m["S"], m["H"], m["W"], seed=m["image_seed"], device="cuda")
    keys = ("depth", "depth_conf", "world_points", "world_points_conf", "part_feat")
    old = mv._HEAD_STREAMS
    try:
        mv._HEAD_STREAMS = "0"
        ref = {k: model(img)[k].clone() for k in keys}
        mv._HEAD_STREAMS = "1"
        for _ in range(4):
            out = model(img)
            torch.cuda.synchronize()
            for k in keys:
                assert torch.equal(out[k], ref[k]), k
        assert model._head_stream is not None
        model.enable_graphs(True)
        for _ in range(3):
            out = model(img)
            torch.cuda.synchronize()
            for k in keys:
                assert torch.equal(out[k], ref[k]), ("graph", k)
    finally:
        mv._HEAD_STREAMS = old
        model.enable_graphs(False)
