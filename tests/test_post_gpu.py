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


@pytest.mark.parametrize("kind", ["surface", "uniform", "ties"])
def test_fill_noise_labels_tiled_equals_brute_force(kind):
    """The local search behind fill_noise_labels on large inputs (sorted pixels, per-tile boxes, far tiles skipped) returns
    exactly what the brute-force kernel returns, ties included (smallest pixel index among equal distances)."""
    g = torch.Generator().manual_seed(11)
    M, C = 60_000, 8
    if kind == "surface":          # points near a 2-D sheet in feature space, like part features of a scene
        uv = torch.rand(M, 2, generator=g)
        basis = torch.randn(2, C, generator=g)
        px = uv @ basis + 0.01 * torch.randn(M, C, generator=g)
    elif kind == "uniform":
        px = torch.rand(M, C, generator=g)
    else:                          # a coarse grid: many exactly equal distances
        px = torch.randint(0, 4, (M, C), generator=g).float()
    labels = torch.randint(0, 7, (M,), generator=g)
    labels[torch.rand(M, generator=g) < 0.15] = -1
    px, labels = px.cuda(), labels.cuda()
    bad = labels < 0
    want = labels.int().clone()
    want[bad] = _C.nn1_label(px[bad].contiguous(), px[~bad].contiguous(), labels[~bad].int().contiguous())
    old = misc.FILL_TILED_MIN
    try:
        misc.FILL_TILED_MIN = 1
        got = misc.fill_noise_labels(px, labels)
    finally:
        misc.FILL_TILED_MIN = old
    assert torch.equal(got, want), int((got != want).sum())


def test_alias_module_exports():
    import iggt.utils.misc as alias

    assert alias.knn_avg_features_pyg is misc.knn_avg_features_pyg and alias.apply_pca_colormap is misc.apply_pca_colormap


# ---- HDBSCAN (csrc/hdbscan.hip + csrc/hdbscan_tree.hip, iggt_official_amd/utils/hdbscan.py) ----------------------------------
def _blobs(rng, n, c, d, spread):
    cen = rng.normal(size=(c, d)) * 3
    return np.concatenate([cen[i] + rng.normal(size=(n // c, d)) * spread * (0.5 + i / c) for i in range(c)]).astype(np.float32)


@pytest.mark.parametrize("M,C,k", [(1000, 8, 10), (5003, 8, 100), (777, 3, 5), (2000, 16, 128), (300, 8, 1)])
def test_hdbscan_core_distances(M, C, k):
    """distance to the k-th nearest row, the row itself counted: sorted brute-force distances in fp64 at column k - 1."""
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, C, generator=g).cuda()
    x[M // 2] = x[M // 3]                                   # a duplicate pair: distance 0 at k = 2
    core = _C.hdbscan_core_dist(x, k)
    ref = torch.sort(torch.cdist(x.double(), x.double()), dim=1).values[:, k - 1]
    assert torch.allclose(core.double(), ref, rtol=2e-6, atol=1e-6), float((core.double() - ref).abs().max())


def test_hdbscan_spanning_tree_and_labels_match_scikit_learn():
    """The whole estimator on the GPU against scikit-learn's (one of the three the reference accepts): spanning-tree weight equal
    to scipy's MST of the dense mutual-reachability matrix, partitions equal (ARI), at the demo's parameters too."""
    from scipy.sparse.csgraph import minimum_spanning_tree
    from scipy.spatial.distance import cdist
    from sklearn.cluster import HDBSCAN
    from sklearn.metrics import adjusted_rand_score

    from iggt_official_amd.utils import hdbscan as hd

    rng = np.random.default_rng(0)
    cases = [
        ("blobs", _blobs(rng, 3000, 5, 8, 0.3), dict(min_samples=10, min_cluster_size=40, eps=0.0)),
        ("blobs eps", _blobs(rng, 3000, 6, 8, 0.4), dict(min_samples=10, min_cluster_size=40, eps=0.5)),
        ("noisy", np.concatenate([_blobs(rng, 2000, 4, 8, 0.3), rng.uniform(-8, 8, size=(600, 8)).astype(np.float32)]),
         dict(min_samples=8, min_cluster_size=30, eps=0.0)),
        ("demo parameters", _blobs(rng, 6000, 4, 8, 0.02), dict(min_samples=100, min_cluster_size=500, eps=0.06)),
        ("3 channels", rng.uniform(size=(1500, 3)).astype(np.float32), dict(min_samples=5, min_cluster_size=20, eps=0.0)),
        ("5 channels (padded)", _blobs(rng, 1200, 3, 5, 0.3), dict(min_samples=5, min_cluster_size=30, eps=0.0)),
    ]
    for name, X, kw in cases:
        x = torch.from_numpy(X).cuda()
        eu, ev, ew, core = hd.mutual_reachability_mst(x, kw["min_samples"])
        D = cdist(X.astype(np.float64), X.astype(np.float64))
        cref = np.sort(D, axis=1)[:, kw["min_samples"] - 1]
        assert np.allclose(core.cpu().numpy(), cref, rtol=3e-6, atol=1e-6), name
        MR = np.maximum(np.maximum(cref[:, None], cref[None, :]), D)
        np.fill_diagonal(MR, 0)
        wref = minimum_spanning_tree(MR).sum()
        assert abs(float(ew.double().sum()) - wref) < 2e-5 * wref, (name, float(ew.double().sum()), wref)
        ref = HDBSCAN(min_samples=kw["min_samples"], min_cluster_size=kw["min_cluster_size"], cluster_selection_epsilon=kw["eps"],
                      algorithm="brute").fit(X.astype(np.float64)).labels_
        got = hd.hdbscan_labels(x, kw["min_cluster_size"], kw["min_samples"], kw["eps"])
        ari = adjusted_rand_score(ref, got)
        from conftest import report
        report(f"post/hdbscan/{name}", dict(points=len(X), clusters=int(got.max() + 1), clusters_ref=int(ref.max() + 1), ari=ari,
                                            noise=int((got < 0).sum()), noise_ref=int((ref < 0).sum())))
        assert got.max() == ref.max() and ari > 0.99, (name, ari)


def test_hdbscan_on_model_features_at_the_demo_scale_matches_scikit_learn():
    """Parity where the estimator is used (reference misc.py:123-129, demo.py:78-83,365-400): 169 344 pixels of the reference
    model's L2-normalised `part_feat` on the demo7 photographs (4 views x 168 x 252, fixture oracle/make_golden_hdbscan.py),
    the demo's parameters (min_samples 100, min_cluster_size 500, epsilon 0.06).  Oracle: scikit-learn's HDBSCAN (kd_tree, 204 s
    on 8 host cores) -- 7 clusters, one of 148 448 pixels and six of 640 ... 994, 15 881 noise pixels.  The GPU estimator must
    find the same partition (ARI >= 0.99, same cluster count; labels agree up to a permutation).  Time and the component count
    after every Boruvka round go to the parity report; the 1.35 M-point timing is test_hdbscan_timing_report."""
    import time

    from sklearn.metrics import adjusted_rand_score

    from conftest import load_golden, report
    from iggt_official_amd.utils import hdbscan as hd

    g = load_golden("hdbscan_demo7_part_feat")
    x = g["features"].float().cuda()
    ref = g["labels"].numpy()
    kw = g["params"]
    hd.hdbscan_labels(x[:20000], kw["min_cluster_size"], kw["min_samples"], kw["cluster_selection_epsilon"])   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = hd.hdbscan_labels(x, kw["min_cluster_size"], kw["min_samples"], kw["cluster_selection_epsilon"])
    dt = time.perf_counter() - t0
    ari = adjusted_rand_score(ref, got)
    both = (ref >= 0) & (got >= 0)
    rep = dict(points=len(ref), clusters=int(got.max() + 1), clusters_ref=int(ref.max() + 1), ari=ari,
               ari_on_pixels_both_label=adjusted_rand_score(ref[both], got[both]), noise=int((got < 0).sum()),
               noise_ref=int((ref < 0).sum()), sizes=sorted(np.bincount(got[got >= 0]).tolist()),
               sizes_ref=sorted(np.bincount(ref[ref >= 0]).tolist()), seconds=dt, seconds_sklearn=g["seconds_sklearn"],
               boruvka_components_per_round=hd.LAST_STATS.get("components_per_round"))
    report("post/hdbscan/demo7_part_feat_169k", rep)
    assert got.max() == ref.max() and ari > 0.99, rep


def test_hdbscan_component_bound_keeps_the_spanning_tree():
    """The per-component pruning of the Boruvka kernel (csrc/hdbscan.hip: workgroups inside one component share the best weight
    found so far) must not change a single edge: same (u, v, w) set with the bound on and off, on clustered and on uniform data,
    including duplicates (equal weights: the index tie-break decides)."""
    from iggt_official_amd.utils import hdbscan as hd

    rng = np.random.default_rng(5)
    dup = _blobs(rng, 4000, 4, 8, 0.05)
    dup[100:1100] = dup[2000:3000]                                     # 1 000 exact duplicates
    cases = [("blobs", _blobs(rng, 20000, 6, 8, 0.05), 20), ("uniform", rng.uniform(size=(9000, 8)).astype(np.float32), 5),
             ("duplicates", dup, 10), ("demo k", _blobs(rng, 30000, 3, 8, 0.02), 100)]
    old = hd.COMPONENT_BOUND
    try:
        for name, X, k in cases:
            x = torch.from_numpy(X).cuda()
            trees = []
            for on in (False, True):
                hd.COMPONENT_BOUND = on
                eu, ev, ew, _ = hd.mutual_reachability_mst(x, k)
                lo, hi = torch.minimum(eu, ev), torch.maximum(eu, ev)
                order = torch.argsort(lo * len(X) + hi)
                trees.append((lo[order].cpu(), hi[order].cpu(), ew[order].cpu()))
            for a, b in zip(trees[0], trees[1]):
                assert torch.equal(a, b), name
    finally:
        hd.COMPONENT_BOUND = old


def test_cluster_features_to_masks_mv_runs_hdbscan_on_the_gpu():
    """reference misc.py:81-170 end to end: HDBSCAN (GPU) -> noise pixels take the nearest labelled pixel's label -> colours.
    Planted, well separated feature clusters on a 3-view map: every pixel ends in its planted cluster."""
    rng = np.random.default_rng(1)
    n, h, w, c = 3, 28, 36, 8
    cen = rng.normal(size=(4, c)).astype(np.float32) * 2
    planted = rng.integers(0, 4, size=(n, h, w))
    fmap = cen[planted] + rng.normal(size=(n, h, w, c)).astype(np.float32) * 0.02
    masks, colored = misc.cluster_features_to_masks_mv(torch.from_numpy(fmap).cuda(), apply_colormap=True, eps=0.06,
                                                       min_samples=20, min_cluster_size=100)
    assert masks.shape == (n, h, w) and colored.shape == (n, h, w, 3) and colored.dtype == np.uint8 and masks.min() >= 0
    from sklearn.metrics import adjusted_rand_score
    assert adjusted_rand_score(planted.reshape(-1), masks.reshape(-1)) == 1.0


def test_hdbscan_timing_report():
    """Wall time of the GPU estimator at 50 k and 200 k points of 8 channels (reported, not gated)."""
    import time

    from conftest import report
    from iggt_official_amd.utils import hdbscan as hd

    rng = np.random.default_rng(2)
    for M in (50_000, 200_000):
        X = _blobs(rng, M, 8, 8, 0.05)
        x = torch.from_numpy(X).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = hd.hdbscan_labels(x, 500, 100, 0.06)
        dt = time.perf_counter() - t0
        report(f"post/hdbscan/timing_{M}", dict(points=M, seconds=dt, clusters=int(got.max() + 1), noise=int((got < 0).sum())))
        assert got.max() + 1 == 8


def test_hdbscan_at_1_35_million_points_is_checked_not_only_timed():
    """Round 5 (review housekeeping): the estimator at the size of the demo's largest call (4 views x 504 x 672 = 1 354 752 pixels,
    8 channels), CHECKED.  scikit-learn cannot run this size in a test (204 s at 169 k points), so the check is made of exact
    properties every minimum spanning tree of the mutual-reachability graph has, verified by brute force on random samples, plus
    the planted partition:
      (a) core distances of 2 048 sampled points = the k-th smallest of their 1.35 M brute-force distances;
      (b) the tree has M - 1 edges, every edge weight equals max(core_u, core_v, |x_u - x_v|) of its endpoints;
      (c) for 2 048 sampled points u: the lightest tree edge at u has the weight of u's lightest mutual-reachability edge to ANY
          other point (brute force over all 1.35 M) -- the lightest edge at every vertex belongs to the tree (cut property), so a
          pruning rule that skipped a tile it should have searched shows up here;
      (d) the tree is connected (scipy's connected_components over its M - 1 edges: one component);
      (e) the labels recover the 8 planted clusters exactly (adjusted Rand index 1 over the non-noise pixels, no cluster missing)."""
    import time

    from conftest import report
    from sklearn.metrics import adjusted_rand_score

    from iggt_official_amd.utils import hdbscan as hd

    rng = np.random.default_rng(5)
    M, k = 1_354_752, 100
    X = _blobs(rng, M, 8, 8, 0.05)
    planted = np.repeat(np.arange(8), M // 8)
    x = torch.from_numpy(X).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eu, ev, ew, core = hd.mutual_reachability_mst(x, k)
    torch.cuda.synchronize()
    t_mst = time.perf_counter() - t0
    assert eu.numel() == M - 1
    # (b)
    d_uv = (x[eu] - x[ev]).double().norm(dim=1)
    mr = torch.maximum(torch.maximum(core[eu], core[ev]).double(), d_uv)
    assert torch.allclose(ew.double(), mr, rtol=3e-6, atol=1e-7), float((ew.double() - mr).abs().max())
    # (a) + (c), brute force in chunks of 256 sampled rows against all points
    g = torch.Generator(device="cuda").manual_seed(7)
    sample = torch.randperm(M, generator=g, device="cuda")[:2048]
    x64 = x.double()
    lightest = torch.full((M,), float("inf"), dtype=torch.float64, device="cuda")
    lightest.scatter_reduce_(0, eu, ew.double(), "amin")
    lightest.scatter_reduce_(0, ev, ew.double(), "amin")
    worst_core, worst_edge = 0.0, 0.0
    for c0 in range(0, sample.numel(), 256):
        idx = sample[c0:c0 + 256]
        D = torch.cdist(x64[idx], x64)                                  # [256, M] exact distances
        kth = torch.topk(D, k, dim=1, largest=False).values[:, k - 1]
        worst_core = max(worst_core, float((core[idx].double() - kth).abs().max() / kth.max()))
        MR = torch.maximum(torch.maximum(kth[:, None], core.double()[None, :]), D)
        MR[torch.arange(idx.numel(), device="cuda"), idx] = float("inf")
        best = MR.amin(1)
        worst_edge = max(worst_edge, float(((lightest[idx] - best).abs() / best).max()))
        del D, MR
    assert worst_core < 3e-6 and worst_edge < 3e-6, (worst_core, worst_edge)
    # (d) connectivity: M - 1 edges spanning one component = a tree
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    eu_h, ev_h = eu.cpu().numpy(), ev.cpu().numpy()
    ncomp, _ = connected_components(coo_matrix((np.ones(M - 1, np.int8), (eu_h, ev_h)), shape=(M, M)), directed=False)
    assert ncomp == 1
    # (e)
    t0 = time.perf_counter()
    got = hd.hdbscan_labels(x, 500, k, 0.06)
    t_all = time.perf_counter() - t0
    keep = got >= 0
    ari = adjusted_rand_score(planted[keep], got[keep])
    report("post/hdbscan/checked_1354752", dict(points=M, mst_seconds=t_mst, labels_seconds=t_all, clusters=int(got.max() + 1),
                                               noise=int((~keep).sum()), ari_vs_planted=ari, core_rel_err_sampled=worst_core,
                                               lightest_edge_rel_err_sampled=worst_edge, rounds=hd.LAST_STATS.get("components_per_round")))
    assert got.max() + 1 == 8 and ari == 1.0 and (~keep).sum() < M // 100
