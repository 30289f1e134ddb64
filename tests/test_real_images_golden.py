"""CPU side of the real-photograph fixtures (BASELINE.json configs[0]; tests/test_real_images_gpu.py is the GPU side).

The fixtures were written by the REFERENCE (oracle/make_golden.py `real`): its own load_fn.load_and_preprocess_images on the
demo JPEGs (torchvision's ToTensor stubbed), its modules, its pose_enc / geometry functions.  Here the CPU restatement
(oracle/restate_utils.py loader, oracle/restate.py model) is pinned against them, so that the oracle the GPU tests lean on at
other sizes is known to hold on photographs too, not only on hash noise.  No GPU needed."""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

CASES = ["real_demo1_s3_crop518_stress", "real_demo7_s4_crop518_stress", "real_demo1_s3_336x504_stress",
         "real_demo7_s4_336x504_stress"]


def _images(m):
    from oracle import restate_utils as ru

    paths = [os.path.join(GOLDEN, "images", m["scene"], f) for f in m["files"]]
    tgt = m["resize_target_size"]
    return ru.load_and_preprocess_images(paths, mode=m["loader_mode"], resize_target_size=None if tgt is None else tuple(tgt))


@pytest.mark.parametrize("case", CASES)
def test_loader_restatement_equals_reference_loader(case):
    g = load_golden(case)
    m = g["meta"]
    images = _images(m)
    assert images.shape == (m["S"], 3, m["H"], m["W"])
    u8 = (images * 255.0).round().to(torch.uint8)
    assert torch.equal(u8.float().div(255), images)
    assert hashlib.sha256(u8.numpy().tobytes()).hexdigest() == m["images_sha256"]
    ss = m["spatial_stride"]
    assert torch.equal(u8[:, :, ::ss, ::ss], g["images_u8_sample"])


def test_model_restatement_matches_reference_on_photographs(schema):
    """demo1 (3 views, 350 x 518 after the 'crop' loader): tokens and every dense output of oracle/restate.py against the
    reference's, plus the camera decode and unprojection chain of demo.py:340-352."""
    from oracle import restate, restate_utils as ru, weights

    case = "real_demo1_s3_crop518_stress"
    g = load_golden(case)
    m = g["meta"]
    ss, ts, cs = m["spatial_stride"], m["token_stride"], m["channel_stride"]
    sd = weights.fill_state_dict(schema, seed=m["weight_seed"], mode=m["mode"])
    with torch.no_grad():
        out = restate.iggt_forward(sd, _images(m))

    def chk(name, got, ref, tol=5e-5):
        d = float((got.double() - ref.double()).abs().max() / ref.double().abs().max())
        assert d < tol, (name, d)

    for li in (4, 11, 17, 23):
        chk(f"tokens_{li}", out["tokens"][li][:, :, ::ts, ::cs], g[f"tokens_{li}"])
    chk("pose_enc", torch.stack(out["pose_enc"], 0), g["pose_enc"])
    for k in ("depth", "depth_conf", "world_points", "world_points_conf"):
        chk(k, out[k][:, :, ::ss, ::ss], g[k])
    assert "part_feat" not in out and "part_feat" not in g          # 350 is not a multiple of 28
    extri, intri = ru.pose_encoding_to_extri_intri(g["pose_enc"][-1], (m["H"], m["W"]))
    assert torch.equal(extri, g["extrinsic"]) and torch.equal(intri, g["intrinsic"])
    world = ru.unproject_depth_map_to_point_map(out["depth"][0].numpy(), extri[0].numpy(), intri[0].numpy())
    chk("world_points_from_depth", torch.from_numpy(np.asarray(world))[:, ::ss, ::ss], g["world_points_from_depth"], tol=1e-4)
