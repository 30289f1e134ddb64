"""HDBSCAN behind cluster_features_to_masks_mv (reference iggt/utils/misc.py:123-129), CPU side.  The oracle is scikit-learn's
HDBSCAN (one of the three estimators the reference accepts, misc.py:19-22).  Checked here without a GPU:
  * the host walk csrc/hdbscan_tree.hip (spanning tree -> dendrogram -> condensed tree -> excess of mass -> epsilon -> labels)
    against the oracle's labels, the spanning tree coming from scipy on the dense mutual-reachability matrix;
  * the Boruvka round logic of iggt_official_amd/utils/hdbscan.py (per-component minimum under the total edge order, hooking,
    pointer jumping, edge de-duplication) with torch re-statements of the two HIP kernels injected: total tree weight equal to
    scipy's, labels equal to the oracle's partition.
The kernels themselves are checked on the GPU by tests/test_post_gpu.py."""
import numpy as np
import pytest
import torch
from scipy.sparse.csgraph import minimum_spanning_tree
from scipy.spatial.distance import cdist
from sklearn.cluster import HDBSCAN
from sklearn.metrics import adjusted_rand_score


def blobs(rng, n, c, d, spread):
    cen = rng.normal(size=(c, d)) * 3
    return np.concatenate([cen[i] + rng.normal(size=(n // c, d)) * spread * (0.5 + i / c) for i in range(c)]).astype(np.float32)


def cases():
    rng = np.random.default_rng(0)
    return [
        ("blobs", blobs(rng, 1500, 5, 8, 0.3), dict(min_samples=10, min_cluster_size=40, eps=0.0)),
        ("blobs eps", blobs(rng, 1500, 6, 8, 0.4), dict(min_samples=10, min_cluster_size=40, eps=0.5)),
        ("blobs eps big", blobs(rng, 1200, 6, 8, 0.4), dict(min_samples=5, min_cluster_size=25, eps=1.5)),
        ("noisy", np.concatenate([blobs(rng, 1000, 4, 8, 0.3), rng.uniform(-8, 8, size=(300, 8)).astype(np.float32)]),
         dict(min_samples=8, min_cluster_size=30, eps=0.0)),
        ("uniform 3-D", rng.uniform(size=(800, 3)).astype(np.float32), dict(min_samples=5, min_cluster_size=20, eps=0.0)),
        ("demo parameters", blobs(rng, 1800, 3, 8, 0.02), dict(min_samples=100, min_cluster_size=500, eps=0.06)),
        ("duplicates", np.repeat(blobs(rng, 300, 3, 8, 0.3), 3, axis=0), dict(min_samples=5, min_cluster_size=30, eps=0.0)),
    ]


def oracle(X, kw, single=False):
    return HDBSCAN(min_samples=kw["min_samples"], min_cluster_size=kw["min_cluster_size"], cluster_selection_epsilon=kw["eps"],
                   allow_single_cluster=single, algorithm="brute").fit(X.astype(np.float64)).labels_


def scipy_mst(X, k):
    D = cdist(X.astype(np.float64), X.astype(np.float64))
    core = np.sort(D, axis=1)[:, k - 1]
    MR = np.maximum(np.maximum(core[:, None], core[None, :]), D)
    np.fill_diagonal(MR, 0)
    T = minimum_spanning_tree(MR).tocoo()
    return T.row.astype(np.int32), T.col.astype(np.int32), T.data.astype(np.float32)


@pytest.mark.parametrize("single", [False, True])
def test_host_walk_matches_the_oracle(single):
    from iggt_official_amd import _C

    for name, X, kw in cases():
        ref = oracle(X, kw, single)
        u, v, w = scipy_mst(X, kw["min_samples"])
        got = _C.hdbscan_labels_from_mst(u, v, w, len(X), kw["min_cluster_size"], kw["eps"], single)
        assert got.max() == ref.max(), (name, got.max(), ref.max())
        # same partition; fp32 edge weights may move a borderline point between a cluster and the noise
        assert adjusted_rand_score(ref, got) > 0.995 and abs(int((got < 0).sum()) - int((ref < 0).sum())) <= 2, name
    with pytest.raises(_C.HipExtensionError):          # not a tree
        _C.hdbscan_labels_from_mst(np.array([0, 0], np.int32), np.array([1, 1], np.int32), np.ones(2, np.float32), 3, 2)
    assert _C.hdbscan_labels_from_mst(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), 1, 2).tolist() == [-1]


def _core_dist_torch(x, k, boxes=None):
    d = torch.cdist(x.double(), x.double())
    return torch.sort(d, dim=1).values[:, k - 1].float()


def _nearest_foreign_torch(xs, core2, comp, idx, tile_lo, tile_hi, boxes=None):
    """Re-statement of hdb_nearest_foreign_kernel: squared mutual reachability in fp32, ties on (min, max) original index."""
    M = xs.shape[0]
    diff = xs[:, None, :] - xs[None, :, :]
    d2 = (diff * diff).sum(-1)
    w = torch.maximum(torch.maximum(core2[:, None], core2[None, :]), d2)
    foreign = comp[:, None] != comp[None, :]
    w = torch.where(foreign, w, torch.full_like(w, float("inf")))
    lo = torch.minimum(idx[:, None], idx[None, :]).long()
    hi = torch.maximum(idx[:, None], idx[None, :]).long()
    wmin = w.amin(1, keepdim=True)
    key = torch.where(w == wmin, lo * M + hi, torch.full_like(lo, 2 ** 62))
    bp = key.argmin(1)
    none = ~foreign.any(1)
    return torch.where(none, torch.full_like(wmin[:, 0], float("inf")), wmin[:, 0]), torch.where(none, torch.full_like(bp, -1), bp).int()


def test_boruvka_round_logic_with_restated_kernels():
    from iggt_official_amd.utils import hdbscan as hd

    kernels = (_core_dist_torch, _nearest_foreign_torch)
    for name, X, kw in cases():
        if len(X) > 1300:
            X = X[::2]
        x = torch.from_numpy(X)
        eu, ev, ew, core = hd.mutual_reachability_mst(x, kw["min_samples"], _kernels=kernels)
        M = len(X)
        assert eu.numel() == M - 1 and int(torch.minimum(eu, ev).min()) >= 0 and int(torch.maximum(eu, ev).max()) < M
        u, v, w = scipy_mst(X, kw["min_samples"])
        assert abs(float(ew.double().sum()) - float(w.astype(np.float64).sum())) < 1e-4 * float(w.sum()), name
        ref = oracle(X, kw)
        got = hd.hdbscan_labels(x, kw["min_cluster_size"], kw["min_samples"], kw["eps"], _kernels=kernels)
        assert adjusted_rand_score(ref, got) > 0.99, (name, adjusted_rand_score(ref, got))
    with pytest.raises(ValueError):
        hd.mutual_reachability_mst(torch.zeros(5, 8), 10, _kernels=kernels)


def test_product_refuses_cpu_points():
    from iggt_official_amd import _C
    from iggt_official_amd.utils import hdbscan as hd

    with pytest.raises(_C.HipExtensionError):
        hd.hdbscan_labels(torch.zeros(50, 8), 5)


def test_plane_folds_match_a_per_position_scan():
    """Host side of the load-balanced searches (iggt_official_amd/_C.py): the per-workgroup planes of a Boruvka round fold under the
    kernel's total order (weight, min original index, max original index); the planes of the split nearest-sample search fold to
    the first minimum.  Random planes with many ties against plain Python scans."""
    from iggt_official_amd import _C

    g = torch.Generator().manual_seed(3)
    G, M = 5, 400
    idx = torch.randperm(M, generator=g).int()
    w2 = torch.randint(0, 4, (G, M), generator=g).float()                 # few distinct weights: ties everywhere
    bp = torch.randint(0, M, (G, M), generator=g).int()
    dead = torch.rand(G, M, generator=g) < 0.3
    w2[dead] = float("inf")
    bp[dead] = -1
    w2[:, 7] = float("inf"); bp[:, 7] = -1                                # a position no plane found anything for
    w, p = _C.fold_foreign_planes(w2, bp, idx)
    for i in range(M):
        cands = [(float(w2[s, i]), min(int(idx[i]), int(idx[bp[s, i]])), max(int(idx[i]), int(idx[bp[s, i]])), int(bp[s, i]))
                 for s in range(G) if bp[s, i] >= 0]
        if not cands:
            assert float(w[i]) == float("inf") and int(p[i]) == -1
            continue
        best = min(cands)
        assert float(w[i]) == best[0]
        j = int(p[i])
        assert (min(int(idx[i]), int(idx[j])), max(int(idx[i]), int(idx[j]))) == best[1:3]
    one_w, one_p = _C.fold_foreign_planes(w2[:1], bp[:1], idx)
    assert torch.equal(one_w, w2[0]) and torch.equal(one_p, bp[0])

    S, Mq, chunk = 6, 300, 50
    d2 = torch.randint(0, 3, (S, Mq), generator=g).float()
    bi = (torch.arange(S)[:, None] * chunk + torch.randint(0, chunk, (S, Mq), generator=g)).int()   # plane s: rows of range s
    empty = torch.rand(S, Mq, generator=g) < 0.2
    d2[empty] = float("inf")
    bi[empty] = -1
    d2[:, 11] = float("inf"); bi[:, 11] = -1
    got = _C.fold_nn1_planes(d2, bi)
    for i in range(Mq):
        cands = [(float(d2[s, i]), int(bi[s, i])) for s in range(S) if bi[s, i] >= 0]
        assert int(got[i]) == (min(cands)[1] if cands else -1)
