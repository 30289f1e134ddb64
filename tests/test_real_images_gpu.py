"""BASELINE.json configs[0] through the HIP path: REAL photographs (GPU).

Every other end-to-end fixture feeds iid hash noise.  White noise gives statistically homogeneous patch tokens -- the best
case for the two devices the headline numbers lean on (mean-input compensation of the fp16 weight rounding, static softmax
bound).  Here the reference's own demo scenes (`iggt_demo/demo1`: 3 photographs 512 x 341, `demo7`: 4 photographs 512 x 512;
copied to tests/golden/images/ as data fixtures) go through the whole caller chain of demo.py:178-202,340-352

    JPEG --load_and_preprocess_images--> [S,3,H,W] --IGGT.forward--> pose_enc / depth / points / part_feat
         --pose_encoding_to_extri_intri--> [R|t], K --unproject_depth_map_to_point_map--> world points

on HIP kernels and are compared with what the REFERENCE produced for the same files (oracle/make_golden.py `real`: the
reference's load_fn.py with torchvision's ToTensor stubbed, its modules with the stress weights, its pose_enc.py /
geometry.py).  Loader: bit-identical (sha256 of the bytes).  Model: the gates of tests/test_e2e_gpu.py.  The token-layer
error is reported with and without mean-input compensation (profiles/r03_parity_report.json, keys real/...)."""
import hashlib
import os

import pytest
import torch

from conftest import GOLDEN, load_golden, report
from helpers import build_gpu_model, errors

pytestmark = pytest.mark.gpu

CASES = ["real_demo1_s3_crop518_stress", "real_demo7_s4_crop518_stress", "real_demo1_s3_336x504_stress",
         "real_demo7_s4_336x504_stress"]
# relative l2 (north_star's 1e-3) and max-abs / range: the gates of tests/test_e2e_gpu.py.  The mean-centred l2 (SURVEY
# section 0 fact 11) gets its own gate on photographs: under the synthetic weights depth = exp(.) varies by only ~1/3 of its
# mean over a photograph (smooth inputs), so ||ref - mean|| is 3x smaller than ||ref|| and the same absolute error reads 3x
# larger -- measured 1.0e-3 ... 1.3e-3 for depth, 1.6e-3 for the points unprojected from it (profiles/r03_parity_report.json)
GATE_L2, GATE_MAX, GATE_L2C = 1e-3, 1.5e-3, 2e-3


@pytest.fixture(autouse=True)
def _restore_precision():
    from iggt_official_amd import precision

    old, old_comp = precision.operand_dtype(), precision._mean_comp
    yield
    precision.set_operand_dtype(old)
    precision.set_mean_compensation(old_comp)


def _paths(m):
    return [os.path.join(GOLDEN, "images", m["scene"], f) for f in m["files"]]


def _load(m):
    from iggt.utils.load_fn import load_and_preprocess_images

    tgt = m["resize_target_size"]
    return load_and_preprocess_images(_paths(m), mode=m["loader_mode"], resize_target_size=None if tgt is None else tuple(tgt))


@pytest.mark.parametrize("case", CASES)
def test_loader_is_bit_identical_to_the_reference_loader(case):
    g = load_golden(case)
    m = g["meta"]
    images = _load(m)
    assert images.is_cuda and images.dtype == torch.float32 and images.shape == (m["S"], 3, m["H"], m["W"])
    images = images.cpu()        # compare on the host: torch's GPU division by a scalar multiplies by the reciprocal
    u8 = (images * 255.0).round().to(torch.uint8)
    assert torch.equal(u8.float().div(255), images)                      # every value is k / 255 exactly (IEEE division)
    ss = m["spatial_stride"]
    assert torch.equal(u8[:, :, ::ss, ::ss], g["images_u8_sample"])
    assert hashlib.sha256(u8.numpy().tobytes()).hexdigest() == m["images_sha256"]


def _forward(model, images):
    cap = {}
    h = model.aggregator.register_forward_hook(lambda mod, i, o: cap.__setitem__("tokens", o[0]))
    pred = model(images)
    h.remove()
    torch.cuda.synchronize()
    return pred, cap["tokens"]


@pytest.mark.parametrize("case", CASES)
def test_demo_chain_matches_reference(case):
    from iggt.utils.geometry import unproject_depth_map_to_point_map
    from iggt.utils.pose_enc import pose_encoding_to_extri_intri
    from iggt_official_amd import precision

    g = load_golden(case)
    m = g["meta"]
    ss, ts, cs = m["spatial_stride"], m["token_stride"], m["channel_stride"]
    model = build_gpu_model(m["mode"], m["weight_seed"])
    images = _load(m)
    precision.set_operand_dtype("f16")
    pred, tokens = _forward(model, images)
    res = {}
    for li in (4, 11, 17, 23):
        res[f"tokens_{li}"] = errors(tokens[li][:, :, ::ts, ::cs], g[f"tokens_{li}"])
    res["pose_enc"] = errors(torch.stack(pred["pose_enc"], 0), g["pose_enc"])
    for k in ("depth", "depth_conf", "world_points", "world_points_conf"):
        res[k] = errors(pred[k][:, :, ::ss, ::ss], g[k])
    if "part_feat" in g:
        res["part_feat"] = errors(pred["part_feat"][:, :, :, ::ss, ::ss], g["part_feat"])
    else:
        assert "part_feat" not in pred
    # the caller's next two steps (demo.py:340-352) on the HIP outputs
    extri, intri = pose_encoding_to_extri_intri(pred["pose_enc"][-1], (m["H"], m["W"]))
    world = unproject_depth_map_to_point_map(pred["depth"][0], extri[0], intri[0], as_tensor=True)
    res["extrinsic"] = errors(extri, g["extrinsic"])
    # a field of view of exactly 0 (the ReLU of the camera head under synthetic weights) gives an infinite focal length in the
    # reference too: same positions, compared apart
    inf_ref = torch.isinf(g["intrinsic"])
    assert torch.equal(torch.isinf(intri).cpu(), inf_ref)
    res["intrinsic"] = errors(torch.where(inf_ref.cuda(), torch.zeros_like(intri), intri),
                              torch.where(inf_ref, torch.zeros_like(g["intrinsic"]), g["intrinsic"]))
    res["world_points_from_depth"] = errors(world[:, ::ss, ::ss], g["world_points_from_depth"])
    # the same forward without mean-input compensation: what the compensation buys on photographs (report only)
    precision.set_mean_compensation(False)
    _, tok_nc = _forward(model, images)
    nocomp = {f"tokens_{li}": errors(tok_nc[li][:, :, ::ts, ::cs], g[f"tokens_{li}"]) for li in (4, 11, 17, 23)}
    precision.set_mean_compensation(True)
    report(f"real/{case}", {k: dict(max=v[0], l2=v[1], l2_centered=v[2]) for k, v in res.items()})
    report(f"real/{case}/static_softmax", model.aggregator.static_softmax_stats())   # flagged query tiles on photographs
    report(f"real/{case}/no_mean_compensation", {k: dict(max=v[0], l2=v[1], l2_centered=v[2]) for k, v in nocomp.items()})
    for k, v in pred.items():
        if torch.is_tensor(v):
            assert torch.isfinite(v).all(), k
    for k, v in res.items():
        assert v[1] < GATE_L2, (k, v)
        if not k.startswith("tokens_"):
            assert v[0] < GATE_MAX and v[2] < GATE_L2C, (k, v)


@pytest.mark.parametrize("case", ["real_demo7_s4_crop518_stress"])
def test_demo_chain_bf16_operands(case):
    """The reference's own GPU arithmetic (autocast bf16, demo.py:190-195) on photographs: the bf16 gates of test_e2e_gpu.py."""
    from iggt_official_amd import precision

    g = load_golden(case)
    m = g["meta"]
    ss, ts, cs = m["spatial_stride"], m["token_stride"], m["channel_stride"]
    model = build_gpu_model(m["mode"], m["weight_seed"])
    images = _load(m)
    precision.set_operand_dtype("bf16")
    pred, tokens = _forward(model, images)
    res = {f"tokens_{li}": errors(tokens[li][:, :, ::ts, ::cs], g[f"tokens_{li}"]) for li in (4, 11, 17, 23)}
    for k in ("depth", "depth_conf", "world_points", "world_points_conf"):
        res[k] = errors(pred[k][:, :, ::ss, ::ss], g[k])
    report(f"real/{case}/bf16", {k: dict(max=v[0], l2=v[1], l2_centered=v[2]) for k, v in res.items()})
    for k, v in res.items():
        assert v[1] < 1e-2 and v[0] < 3e-2, (k, v)


@pytest.mark.parametrize("case", ["real_demo7_s4_crop518_stress", "real_demo7_s4_336x504_stress"])
def test_head_convolutions_two_passes_vs_three_on_photographs(case):
    """The DPT heads at convops.DPT_PREC = 2 (fp16 hi + lo activations x fp16 weights + mean-input compensation: two MFMA
    passes, what ships) against the same forward at 3 (split-bf16, fp32-grade): the head-only difference the cheaper operand
    format costs -- VERDICT r2 item 5 allows 5e-4 l2 / 1e-3 max per head output (probes/conv_precision.py predicted 1.7e-4) --
    and both against the reference fixture."""
    from iggt_official_amd import precision
    from iggt_official_amd.heads import convops as co

    g = load_golden(case)
    m = g["meta"]
    ss = m["spatial_stride"]
    model = build_gpu_model(m["mode"], m["weight_seed"])
    images = _load(m)
    precision.set_operand_dtype("f16")
    old = co.DPT_PREC
    try:
        def run():   # pose_enc is a list (one entry per refinement iteration): keep the final one
            return {k: (v[-1] if isinstance(v, (list, tuple)) else v).clone() for k, v in model(images).items()
                    if torch.is_tensor(v) or isinstance(v, (list, tuple))}

        co.DPT_PREC = 3
        p3 = run()
        co.DPT_PREC = 2
        p2 = run()
    finally:
        co.DPT_PREC = old
    out = {}
    for k in ("depth", "depth_conf", "world_points", "world_points_conf"):
        d = errors(p2[k], p3[k])
        out[k] = dict(head_only_max=d[0], head_only_l2=d[1],
                      prec3_vs_reference_l2=errors(p3[k][:, :, ::ss, ::ss], g[k])[1],
                      prec2_vs_reference_l2=errors(p2[k][:, :, ::ss, ::ss], g[k])[1])
        assert d[1] < 5e-4 and d[0] < 1e-3, (k, d)
    assert torch.equal(p2["pose_enc"], p3["pose_enc"])       # the camera head has no convolution
    report(f"real/{case}/head_conv_prec2_vs_prec3", out)
