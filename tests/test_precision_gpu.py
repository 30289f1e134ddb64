"""fp16 operand-format safety nets (GPU): the debug saturation counter, outlier channels of the kind real DINOv2 / VGGT
checkpoints carry ("massive activations"), weight-range validation at pack time and pack invalidation on load_state_dict."""
import pytest
import torch

from helpers import build_gpu_model, errors, schema

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset():
    from iggt_official_amd import precision

    yield
    precision.set_debug_saturation(False)


def _tiny_inputs():
    from oracle import weights

    return weights.make_images(2, 56, 56, seed=1, device="cuda")


def test_saturation_counter_and_outlier_channels():
    """(a) stress weights: nothing saturates.  (b) four residual-stream channels 200x larger than the rest (camera / register
    tokens and a LayerScale row): still nothing saturates and the outputs stay within 1e-3 of the CPU fp32 restatement of the
    reference.  (c) an MLP weight row scaled until the hidden activation leaves the fp16 range: the counter reports it."""
    from iggt_official_amd import precision
    from oracle import restate, weights

    model = build_gpu_model("stress", 0)
    images = _tiny_inputs()
    precision.set_debug_saturation(True)
    model(images)
    rep = precision.saturation_report()
    assert rep and all(v == 0 for v in rep.values()), rep

    sd = weights.fill_state_dict(schema(), seed=0, mode="stress", device="cuda")
    sd["aggregator.register_token"][..., :4] *= 200.0
    sd["aggregator.camera_token"][..., 7:9] *= 200.0
    sd["aggregator.frame_blocks.2.ls2.gamma"][100:104] *= 30.0
    try:
        model.load_state_dict(sd, strict=False)
        precision.set_debug_saturation(True)
        pred = model(images)
        torch.cuda.synchronize()
        rep = precision.saturation_report()
        assert all(v == 0 for v in rep.values()), rep
        cpu_sd = {k: v.cpu() for k, v in sd.items()}
        with torch.no_grad():
            ref = restate.iggt_forward(cpu_sd, images.cpu(), with_part=True)
        for k in ("depth", "world_points", "part_feat"):
            e = errors(pred[k], ref[k])
            assert e[1] < 1e-3, (k, e)

        sd["aggregator.global_blocks.1.mlp.fc1.weight"][:8] *= 2.0e5      # hidden activations of 8 units ~ 2e5 > 65504 (weights themselves stay < 65504)
        model.load_state_dict(sd, strict=False)
        precision.set_debug_saturation(True)
        model(images)
        rep = precision.saturation_report()
        assert rep["mlp_hidden"] > 0, rep
    finally:
        model.load_state_dict(weights.fill_state_dict(schema(), seed=0, mode="stress", device="cuda"), strict=False)


def test_weight_beyond_fp16_range_without_a_fold_partner():
    """One weight ENTRY of 1e5 is not a re-parametrisation (its row / column medians are ordinary), so nothing can absorb it:
    with range folding (the default) that one block runs on bf16 operands and every other block stays on fp16; with folding
    off the checkpoint is rejected at pack time as in round 3, and bf16 operands for the whole trunk remain the way out."""
    from iggt_official_amd import precision
    from oracle import weights

    model = build_gpu_model("stress", 0)
    sd = weights.fill_state_dict(schema(), seed=0, mode="stress", device="cuda")
    good = sd["aggregator.frame_blocks.0.attn.proj.weight"].clone()
    sd["aggregator.frame_blocks.0.attn.proj.weight"][3, 5] = 1.0e5
    try:
        model.load_state_dict(sd, strict=False)
        out = model(_tiny_inputs())
        assert torch.isfinite(out["depth"]).all() and precision.operand_name() == "f16"
        blocks = model.aggregator.frame_blocks
        assert blocks[0].packed()["w_qkv"].dtype == torch.bfloat16 and blocks[1].packed()["w_qkv"].dtype == torch.float16
        precision.set_range_folding(False)
        try:
            with pytest.raises(ValueError, match="fp16"):
                model(_tiny_inputs())
            old = precision.operand_dtype()
            try:
                precision.set_operand_dtype(torch.bfloat16)        # the documented way out: bf16 operands have fp32's range
                out = model(_tiny_inputs())
                assert torch.isfinite(out["depth"]).all()
            finally:
                precision.set_operand_dtype(old)
        finally:
            precision.set_range_folding(True)
    finally:
        sd["aggregator.frame_blocks.0.attn.proj.weight"] = good
        model.load_state_dict(sd, strict=False)


def test_range_folding_keeps_fp16_operands_for_reparametrised_checkpoints():
    """The same function written with weight columns / rows of magnitude ~1e5 (a LayerNorm scale of 1e-5 in front of a 1e5
    qkv / fc1 column, a LayerScale of 1e-5 behind a 1e5 proj / fc2 row, a tiny V channel in front of a 1e5 proj column --
    every place where a power of two (here 2^22) moves between two partners exactly): the checkpoint does not fit fp16 as stored, range
    folding (layers/blocks.py fold_ranges) undoes the re-parametrisation at pack time, the trunk stays on fp16 operands, no
    16-bit store saturates and the outputs stay within 1e-3 of the CPU fp32 restatement run on the checkpoint AS STORED."""
    from iggt_official_amd import precision
    from oracle import restate, weights

    K = 2.0 ** 22          # weights of ~0.05 become ~2e5: beyond the fp16 range
    model = build_gpu_model("stress", 0)
    images = _tiny_inputs()
    sd = weights.fill_state_dict(schema(), seed=0, mode="stress", device="cuda")
    for kind, i in (("frame_blocks", 0), ("global_blocks", 3), ("frame_blocks", 11), ("global_blocks", 23)):
        p = f"aggregator.{kind}.{i}."
        sd[p + "attn.qkv.weight"][:, 5] *= K
        sd[p + "norm1.weight"][5] /= K
        sd[p + "norm1.bias"][5] /= K
        sd[p + "attn.proj.weight"][7] *= K
        sd[p + "attn.proj.bias"][7] *= K
        sd[p + "ls1.gamma"][7] /= K
        sd[p + "attn.proj.weight"][:, 9] *= K
        sd[p + "attn.qkv.weight"][2048 + 9] /= K
        sd[p + "attn.qkv.bias"][2048 + 9] /= K
        sd[p + "mlp.fc1.weight"][:, 3] *= K
        sd[p + "norm2.weight"][3] /= K
        sd[p + "norm2.bias"][3] /= K
        sd[p + "mlp.fc2.weight"][11] *= K
        sd[p + "mlp.fc2.bias"][11] *= K
        sd[p + "ls2.gamma"][11] /= K
    assert float(sd["aggregator.global_blocks.3.attn.qkv.weight"].abs().max()) > 65504.0
    try:
        model.load_state_dict(sd, strict=False)
        precision.set_debug_saturation(True)
        pred = model(images)
        torch.cuda.synchronize()
        rep = precision.saturation_report()
        assert precision.operand_name() == "f16"
        assert all(v == 0 for v in rep.values()), rep
        pk = model.aggregator.global_blocks[3].packed()
        assert pk["w_qkv"].dtype == torch.float16 and pk["folded_slices"] == 5
        assert model.aggregator.global_blocks[4].packed()["folded_slices"] == 0
        cpu_sd = {k: v.cpu() for k, v in sd.items()}
        with torch.no_grad():
            ref = restate.iggt_forward(cpu_sd, images.cpu(), with_part=True)
        for k in ("depth", "world_points", "part_feat"):
            e = errors(pred[k], ref[k])
            assert e[1] < 1e-3, (k, e)
    finally:
        model.load_state_dict(weights.fill_state_dict(schema(), seed=0, mode="stress", device="cuda"), strict=False)


def test_load_state_dict_after_forward_rebuilds_packs():
    """Packed 16-bit weights, dW copies, conv hi/lo packs and BN folds are keyed on the parameters' version: after a forward,
    load_state_dict with different values must give exactly what a freshly built model gives."""
    from oracle import weights

    images = _tiny_inputs()
    model = build_gpu_model("stress", 0)
    a0 = model(images)["world_points"].clone()
    model.load_state_dict(weights.fill_state_dict(schema(), seed=1, mode="stress", device="cuda"), strict=False)
    b = {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in model(images).items()}
    assert not torch.equal(b["world_points"], a0)
    fresh = build_gpu_model("stress", 1)           # a new model object with the seed-1 weights
    c = fresh(images)
    for k in ("depth", "depth_conf", "world_points", "world_points_conf", "part_feat"):
        assert torch.equal(b[k], c[k]), k
    assert all(torch.equal(x, y) for x, y in zip(b["pose_enc"], c["pose_enc"]))


def test_large_slice_with_an_ordinary_partner_is_not_folded():
    """a qkv input column that is genuinely 2^22 x larger -- its LayerNorm scale NOT correspondingly small -- is not a
    re-parametrisation.  Folding it would hand the factor to norm1 (scale 4e6), the fp16 LayerNorm output would saturate and the
    GEMM turn it into inf.  The fold is refused (partner bound, layers/blocks.py PARTNER_MAX): the block runs on bf16 operands as
    an un-foldable one does, the others stay on fp16, and every output is finite.  NaN weights raise instead of choosing a format."""
    from iggt_official_amd import precision
    from oracle import weights

    model = build_gpu_model("stress", 0)
    sd = weights.fill_state_dict(schema(), seed=0, mode="stress", device="cuda")
    key = "aggregator.global_blocks.2.attn.qkv.weight"
    good = sd[key].clone()
    try:
        sd[key][:, 5] *= 2.0 ** 22
        model.load_state_dict(sd, strict=False)
        out = model(_tiny_inputs())
        torch.cuda.synchronize()
        for k, v in out.items():
            if torch.is_tensor(v):
                assert torch.isfinite(v).all(), k
        pk = model.aggregator.global_blocks[2].packed()
        assert pk["folded_slices"] == 0 and pk["w_qkv"].dtype == torch.bfloat16 and pk["bf16_fallback"]
        assert model.aggregator.global_blocks[3].packed()["w_qkv"].dtype == torch.float16
        assert "global_blocks.2" in model.aggregator.escalation_report()["bf16_fallback"]
        sd[key] = good.clone()
        sd[key][0, 0] = float("nan")
        model.load_state_dict(sd, strict=False)
        with pytest.raises(ValueError, match="NaN"):
            model(_tiny_inputs())
    finally:
        sd[key] = good
        model.load_state_dict(sd, strict=False)
