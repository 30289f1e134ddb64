"""Pin the CPU restatement (oracle/restate.py) against the golden fixtures produced by the REFERENCE
modules (oracle/make_golden.py): same seeded weights/inputs, fp32 CPU both sides.  Tolerance 2e-5
relative to the tensor's max (pure fp32 re-association differences)."""
import pytest
import torch

from conftest import load_golden

CASES = ["tiny_s2_56_stress", "tiny_s3_84x56_stress", "tiny_s2_70_stress"]
TOL = 5e-5


@pytest.fixture(scope="module")
def sd(schema):
    from oracle import weights

    return weights.fill_state_dict(schema, seed=0, mode="stress")


def _chk(name, got, ref, tol=TOL):
    d = float((got.double() - ref.double()).abs().max() / ref.double().abs().max())
    assert d < tol, (name, d)


@pytest.mark.parametrize("case", CASES)
def test_restatement_matches_reference_outputs(sd, case):
    from oracle import restate, weights

    g = load_golden(case)
    m = g["meta"]
    assert m["mode"] == "stress" and m["weight_seed"] == 0
    images = weights.make_images(m["S"], m["H"], m["W"], seed=m["image_seed"])
    out = restate.iggt_forward(sd, images)
    for li in (4, 11, 17, 23):
        _chk(f"tokens_{li}", out["tokens"][li], g[f"tokens_{li}"])
    _chk("pose_enc", torch.stack(out["pose_enc"], 0), g["pose_enc"])
    for k in ("depth", "depth_conf", "world_points", "world_points_conf"):
        _chk(k, out[k], g[k])
    for i in range(3):
        _chk(f"point_feat_{i}", out["point_feat"][i], g[f"point_feat_{i}"])
    if "part_feat" in g:
        for i, k in enumerate(("res1", "res2", "res3", "res4")):
            _chk(f"adaptor_{k}", out["adaptor"][i], g[f"adaptor_{k}"])
        _chk("part_feat", out["part_feat"], g["part_feat"])
    else:
        assert "part_feat" not in out


def test_golden_weights_are_reproducible(schema):
    """The synthetic weights are a pure function of (name, seed, mode): spot-check known values."""
    from oracle import weights

    w = weights.make_tensor("aggregator.frame_blocks.3.ls1.gamma", (1024,), 0, "stress")
    assert torch.allclose(w[:4], torch.tensor([0.5007, 0.9852, 0.8057, 0.6321]), atol=1e-4)
    a = weights.make_tensor("depth_head.scratch.output_conv1.weight", (128, 256, 3, 3), 0, "stress")
    b = weights.make_tensor("depth_head.scratch.output_conv1.weight", (128, 256, 3, 3), 0, "stress")
    assert torch.equal(a, b) and abs(float(a.std()) * (256 * 9) ** 0.5 - 1.0) < 0.01


def test_golden_recipe_imports_the_reference_and_reproduces_a_fixture(tmp_path):
    """`python oracle/make_golden.py` must (a) resolve `iggt` to the REFERENCE checkout although this repository ships
    a regular `iggt/` alias package of the product, and (b) reproduce a committed fixture bit for bit.  Runs in a
    child process (the shim rebinds `iggt` in sys.modules).  Skipped where /root/reference does not exist."""
    import os
    import subprocess
    import sys

    from conftest import GOLDEN, ROOT
    from oracle import ref_shim

    if not ref_shim.available():
        pytest.skip("no reference checkout on this machine")
    probe = ("import sys; sys.path.insert(0, %r); import iggt.models.vggt as prod; "
             "from oracle import ref_shim; ref_shim.install(); import iggt.models.vggt as v, iggt.heads.dpt_head as d; "
             "print(prod.__file__); print(v.__file__); print(d.__file__)" % ROOT)
    out = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    prod, ref_v, ref_d = out.stdout.strip().splitlines()[-3:]
    assert prod.startswith(ROOT) and ref_v.startswith(ref_shim.REF_ROOT) and ref_d.startswith(ref_shim.REF_ROOT)

    case = "tiny_s2_56_stress"
    run = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py"), "--out", str(tmp_path), case],
                         cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert run.returncode == 0, run.stderr[-2000:]
    new = torch.load(os.path.join(str(tmp_path), case + ".pt"), weights_only=False)
    old = torch.load(os.path.join(GOLDEN, case + ".pt"), weights_only=False)
    assert set(new) == set(old)
    for k, v in old.items():
        if torch.is_tensor(v):
            assert torch.equal(new[k], v), k


# ------------------------------------------------------------------------------------------------
# track head (query_points path): restatement oracle/restate_track.py against the reference's own TrackHead outputs
TRACK_CASES = ["track_s3_140_stress", "track_s2_140x182_stress"]


@pytest.fixture(scope="module")
def sd_track(schema):
    from oracle import weights

    return weights.fill_state_dict(schema, seed=0, mode="stress", include_track=True)


@pytest.mark.parametrize("case", TRACK_CASES)
def test_track_restatement_matches_reference(sd_track, case):
    from oracle import restate, restate_track, weights

    g = load_golden(case)
    m = g["meta"]
    images = weights.make_images(m["S"], m["H"], m["W"], seed=m["image_seed"])
    toks = restate.aggregator(sd_track, images)
    taps = {}
    preds, vis, conf = restate_track.track_head(sd_track, toks, m["H"], m["W"], g["query_points"], taps=taps)
    _chk("fmaps", taps["fmaps"], g["fmaps"])
    _chk("fcorrs_it0", taps["fcorrs_it0"], g["fcorrs_it0"])
    _chk("delta_it0", taps["delta_it0"], g["delta_it0"], tol=2e-4)
    # the flow embedding multiplies coordinate differences by up to 969 rad / pixel (utils.py:107): fp32 re-association
    # noise of the first iteration (1e-6 pixels) comes back as 1e-3 rad phase noise in the second
    _chk("coord_preds", torch.stack(preds, 0), g["coord_preds"], tol=2e-4)
    _chk("vis", vis, g["vis"], tol=2e-3)
    _chk("conf", conf, g["conf"], tol=2e-3)
    # the tracker alone, fed with the reference's feature maps
    preds2, vis2, conf2 = restate_track.tracker(sd_track, g["fmaps"], g["query_points"])
    _chk("coord_preds (reference fmaps)", torch.stack(preds2, 0), g["coord_preds"], tol=2e-4)
