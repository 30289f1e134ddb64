"""GPU: the post-processing kernels (csrc/postprocess.hip via iggt_official_amd/utils/misc.py) against the reference fixture
and against the CPU restatement: neighbour SETS bit-exact, means / colours to fp32 accuracy, labels and colours bit-exact."""
import os

import numpy as np
import pytest
import torch

from iggt_official_amd import _C
from iggt_official_amd.utils import misc
from oracle import restate_post
from test_post_golden import GOLD, flip_aware_max_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, weights_only=False)


def test_knn_average_matches_reference_fixture(gold):
    out = misc.knn_avg_features_pyg(gold["points"], gold["features"], gold["k"])
    assert out.is_cuda and out.shape == gold["knn_avg"].shape
    assert torch.allclose(out.cpu(), gold["knn_avg"], rtol=0, atol=2e-6)   # same neighbour sets, fp32 sums in another order


@pytest.mark.parametrize("M,k,kind", [(1, 5, "gauss"), (7, 20, "gauss"), (300, 8, "gauss"), (5000, 20, "gauss"),
                                      (20000, 20, "surface"), (20000, 32, "line"), (9000, 16, "outliers"),
                                      (4097, 1, "gauss")])
def test_knn_sets_are_exact(M, k, kind):
    g = torch.Generator().manual_seed(M + k)
    p = torch.randn(M, 3, generator=g)
    if kind == "surface":
        p[:, 2] = 0.3 * torch.sin(3 * p[:, 0]) + 1e-3 * p[:, 2]
    elif kind == "line":                       # degenerate: no pruning along two axes
        p[:, 1:] = 0
    elif kind == "outliers":
        p[::97] *= 1e4
        p[5] = float("inf")                    # a non-finite point has no neighbours and is nobody's neighbour
    idx, d2 = misc.knn_indices(p.cuda(), k, return_sq_dist=True)
    idx, d2 = idx.cpu().long(), d2.cpu()
    if M == 1:
        assert (idx == -1).all()
        return
    finite = torch.isfinite(p).all(1)
    ref = restate_post.knn_index_sets(p[finite], k)                       # indices into the finite subset
    remap = torch.nonzero(finite)[:, 0]
    ref = remap[ref]
    kk = ref.shape[1]
    got = torch.sort(idx[finite][:, :kk], dim=1).values
    want = torch.sort(ref, dim=1).values
    # bit-exact neighbour sets, except where the k-th and (k+1)-th squared distances agree to fp32 rounding (the kernel
    # measures in fp32, the restatement in fp64): there either choice is a correct answer
    pf = p[finite].double()
    for r in torch.nonzero((got != want).any(1))[:, 0].tolist():
        a, b = set(got[r].tolist()), set(want[r].tolist())
        dk = float(((pf[r] - p[sorted(b)].double()) ** 2).sum(-1).max())
        for j in a ^ b:
            assert abs(float(((pf[r] - p[j].double()) ** 2).sum()) - dk) <= 1e-6 * dk, (r, j)
    assert (idx[finite][:, kk:] == -1).all()
    assert (idx[~finite] == -1).all()
    # ascending distances, equal to the exact squared distances
    dd = d2[finite][:, :kk]
    assert (dd[:, 1:] >= dd[:, :-1]).all()
    exact = ((p[finite][:, None, :] - p[idx[finite][:, :kk]]) ** 2).sum(-1)
    assert torch.allclose(dd, exact, rtol=1e-5, atol=0)


def test_pca_colormap_matches_reference_fixture(gold):
    for src, key in ((gold["features"], "pca_raw"), (gold["knn_avg"], "pca_smooth")):
        out = misc.apply_pca_colormap(src.cuda())
        assert out.shape == gold[key].shape and float(out.min()) >= 0 and float(out.max()) <= 1
        assert flip_aware_max_err(out.cpu(), gold[key]) < 2e-4
    # and against the exact-SVD restatement with the orientation convention shared: same colours
    img = gold["features"]
    axes = misc.pca_axes(img.reshape(-1, img.shape[-1]).cuda().contiguous()).cpu()
    ours = misc.apply_pca_colormap(img.cuda()).cpu()
    assert float((ours - restate_post.pca_colormap(img, sign_like=axes)).abs().max()) < 1e-4


def test_pca_colormap_constant_channel_and_offset():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 16, 16, 2, generator=g) * torch.tensor([3.0, 0.5])
    x = torch.cat([x, torch.zeros(2, 16, 16, 3)], -1) + 100.0            # rank 2, large common offset
    out = misc.apply_pca_colormap(x.cuda()).cpu()
    ref = restate_post.pca_colormap(x, sign_like=misc.pca_axes(x.reshape(-1, 5).cuda().contiguous()).cpu())
    assert float((out[..., :2] - ref[..., :2]).abs().max()) < 2e-3          # offset 100 costs fp32 digits in the projection


def test_cluster_masks_match_reference_fixture(gold):
    planted = gold["planted_labels"].numpy()
    masks, colored = misc.cluster_features_to_masks_mv(gold["knn_avg"].cuda(), apply_colormap=True,
                                                       clusterer=lambda px: planted, eps=0.06, min_samples=100,
                                                       min_cluster_size=500)
    assert masks.dtype == np.int64 and colored.dtype == np.uint8
    assert np.array_equal(masks, gold["masks"].numpy())
    assert np.array_equal(colored, gold["colored"].numpy())
    only = misc.cluster_features_to_masks_mv(gold["knn_avg"], clusterer=lambda px: planted)
    assert np.array_equal(only, masks)
    allnoise = misc.cluster_features_to_masks_mv(gold["knn_avg"], clusterer=lambda px: np.full(px.shape[0], -1))
    assert (allnoise == 0).all()


def test_cluster_with_host_hdbscan_runs():
    """HDBSCAN itself is a host library call (scikit-learn here): two well separated blobs -> two labels, no noise left."""
    g = torch.Generator().manual_seed(0)
    a = torch.randn(1, 20, 20, 8, generator=g) * 0.02
    a[:, :, 10:] += 1.0
    masks = misc.cluster_features_to_masks_mv(a.cuda(), eps=0.06, min_samples=10, min_cluster_size=50)
    assert masks.shape == (1, 20, 20) and masks.min() >= 0
    assert len(np.unique(masks[:, :, :10])) == 1 and len(np.unique(masks[:, :, 10:])) == 1
    assert masks[0, 0, 0] != masks[0, 0, 19]


def test_nn1_label_first_minimum():
    g = torch.Generator().manual_seed(2)
    ref = torch.randn(3000, 8, generator=g)
    q = torch.randn(1000, 8, generator=g)
    lab = torch.arange(3000, dtype=torch.int32)
    got = _C.nn1_label(q.cuda(), ref.cuda(), lab.cuda()).cpu().long()
    want = torch.cdist(q.double(), ref.double()).argmin(1)
    assert torch.equal(got, want)


def test_alias_module_exports():
    import iggt.utils.misc as alias

    assert alias.knn_avg_features_pyg is misc.knn_avg_features_pyg and alias.apply_pca_colormap is misc.apply_pca_colormap
