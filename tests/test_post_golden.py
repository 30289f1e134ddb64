"""CPU: the restatement of the demo post-processing step (oracle/restate_post.py) against the fixture the reference functions
wrote (oracle/make_golden_post.py -> tests/golden/post_misc.pt)."""
import os

import numpy as np
import pytest
import torch

from oracle import restate_post

GOLD = os.path.join(os.path.dirname(__file__), "golden", "post_misc.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, weights_only=False)


def flip_aware_max_err(ours, ref):
    """apply_pca_colormap is defined up to the sign of each principal axis: a flipped axis turns channel c into 1 - c."""
    errs = []
    for ch in range(3):
        a, b = ours[..., ch], ref[..., ch]
        errs.append(min(float((a - b).abs().max()), float((1 - a - b).abs().max())))
    return max(errs)


def test_knn_average_restatement(gold):
    out = restate_post.knn_avg_features(gold["points"], gold["features"], gold["k"])
    assert torch.allclose(out, gold["knn_avg"], rtol=0, atol=1e-6)


def test_pca_colormap_restatement(gold):
    for src, key in ((gold["features"], "pca_raw"), (gold["knn_avg"], "pca_smooth")):
        out = restate_post.pca_colormap(src)
        assert out.shape == gold[key].shape
        # exact SVD vs the reference's randomised range finder at full rank: same axes to fp32 accuracy
        assert flip_aware_max_err(out, gold[key]) < 2e-4


def test_label_fill_restatement(gold):
    flat = gold["knn_avg"].reshape(-1, gold["knn_avg"].shape[-1]).numpy()
    lab = restate_post.fill_noise_labels(flat, gold["planted_labels"].numpy())
    assert np.array_equal(lab.reshape(gold["masks"].shape), gold["masks"].numpy())
    assert restate_post.fill_noise_labels(flat, np.full(flat.shape[0], -1)).sum() == 0
