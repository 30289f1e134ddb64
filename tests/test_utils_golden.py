"""Pin the CPU restatement of the pose / geometry utilities (oracle/restate_utils.py) against golden vectors produced by the
REFERENCE functions (oracle/make_golden_utils.py -> tests/golden/utils_pose_geometry.pt), and the host-side logic of the GPU
image loader (coefficient tables of Pillow's resampler) against PIL itself.  No GPU needed."""
import numpy as np
import pytest
import torch

from conftest import load_golden


@pytest.fixture(scope="module")
def g():
    return load_golden("utils_pose_geometry")


def test_pose_decode_restatement_matches_reference(g):
    from oracle import restate_utils as ru

    extri, intri = ru.pose_encoding_to_extri_intri(g["pose"], (g["H"], g["W"]))
    assert torch.equal(extri, g["extri"]) and torch.equal(intri, g["intri"])   # same fp32 ops in the same order


def test_unprojection_restatement_matches_reference(g):
    from oracle import restate_utils as ru

    world = ru.unproject_depth_map_to_point_map(g["depth"].numpy(), g["extri"][0].numpy(), g["intri"][0].numpy())
    assert world.dtype == np.float64 and np.array_equal(world, g["world"].numpy())
    assert np.array_equal(world[1], g["frame1_world"].numpy())


def test_closed_form_inverse_se3_matches_reference(g):
    from iggt_official_amd.utils.geometry import closed_form_inverse_se3

    inv = closed_form_inverse_se3(g["extri"][0].numpy())
    assert np.allclose(inv, g["inv_se3"].numpy(), rtol=0, atol=1e-7)
    inv_t = closed_form_inverse_se3(g["extri"][0])
    assert torch.allclose(inv_t.double(), g["inv_se3"], atol=1e-6)


def test_resampler_tables_reproduce_pillow_bit_for_bit():
    """The coefficient tables the GPU loader uploads (iggt_official_amd/utils/load_fn._coeffs), applied with the kernels'
    integer arithmetic in numpy, give exactly PIL.Image.resize(BICUBIC) -- up-scaling, down-scaling and mixed."""
    from PIL import Image

    from iggt_official_amd.utils.load_fn import _coeffs

    def emulate(img, new_w, new_h):
        a = np.asarray(img).astype(np.int64)
        h, w, _ = a.shape
        (hb, hk), (vb, vk) = _coeffs(w, new_w), _coeffs(h, new_h)
        hb, hk, vb, vk = (t.numpy().astype(np.int64) for t in (hb, hk, vb, vk))
        tmp = np.zeros((h, new_w, 3), np.int64)
        for xx in range(new_w):
            x0, n = hb[xx]
            tmp[:, xx] = np.clip(((1 << 21) + (a[:, x0:x0 + n] * hk[xx, :n][None, :, None]).sum(1)) >> 22, 0, 255)
        out = np.zeros((new_h, new_w, 3), np.int64)
        for yy in range(new_h):
            y0, n = vb[yy]
            out[yy] = np.clip(((1 << 21) + (tmp[y0:y0 + n] * vk[yy, :n][:, None, None]).sum(0)) >> 22, 0, 255)
        return out.astype(np.uint8)

    rng = np.random.default_rng(0)
    for (h, w), (nw, nh) in (((341, 512), (518, 350)), ((600, 800), (518, 392)), ((90, 100), (518, 574)),
                             ((1200, 700), (308, 518))):
        img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        ref = np.asarray(img.resize((nw, nh), Image.Resampling.BICUBIC))
        assert np.array_equal(emulate(img, nw, nh), ref), ((h, w), (nw, nh))


def test_align_and_update_state_dicts_contract():
    """reference utils/model.py:27-55: matched = same key and shape; everything else is reported, not loaded."""
    import logging

    from utils.model import align_and_update_state_dicts

    msgs = []

    class L(logging.Logger):
        def warning(self, m, *a, **k):
            msgs.append(m)

        def info(self, m, *a, **k):
            pass

    model_sd = {"a.weight": torch.zeros(4, 3), "b.bias": torch.zeros(5), "c.weight": torch.zeros(2, 2)}
    ckpt = {"a.weight": torch.ones(4, 3), "b.bias": torch.ones(6), "extra": torch.ones(1)}
    out = align_and_update_state_dicts(L("t"), model_sd, ckpt)
    assert list(out) == ["a.weight"] and torch.equal(out["a.weight"], ckpt["a.weight"])
    joined = "\n".join(msgs)
    assert "*UNMATCHED* b.bias" in joined and "*UNLOADED* c.weight" in joined and "$UNUSED$ extra" in joined
