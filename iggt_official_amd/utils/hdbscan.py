"""HDBSCAN on the GPU -- the clustering step of `cluster_features_to_masks_mv` (reference iggt/utils/misc.py:123-129:
`HDBSCAN(cluster_selection_epsilon=eps, min_samples=..., min_cluster_size=..., allow_single_cluster=False).fit(pixels).labels_`
with cuml's / hdbscan's / scikit-learn's estimator, misc.py:19-22; caller demo.py:385-394 on all N * H * W pixels' part features).

The algorithm (Campello, Moulavi, Sander 2013; cluster_selection_epsilon: Malzer & Baum 2020), exact, in three stages:
  1. core distances: distance to the min_samples-th nearest point, itself counted       -- csrc/hdbscan.hip, brute force, fp32
  2. minimum spanning tree of the mutual-reachability graph max(core_i, core_j, d_ij), by Boruvka rounds: every round one kernel
     finds, for every point, the cheapest edge into another component (edges totally ordered by (weight, min index, max index));
     the per-component minimum (segmented reductions over the component-sorted arrays), the hooking of components and the pointer
     jumping are torch index operations on M-element arrays
  Both kernels are exhaustive searches made local: the points are first ordered along a Morton curve over their first three
  principal axes (`spatial_order`), every 256-point tile carries its bounding box, and a tile that is provably too far is skipped.
  A round only needs the cheapest outgoing edge per COMPONENT: workgroups inside one component share the best weight found so far
  and drop everything strictly worse (exact; interior points of a large component stop after their first foreign tile).
  3. dendrogram -> condensed tree -> excess-of-mass selection -> epsilon -> labels: a walk over the M - 1 edges on the host
     (csrc/hdbscan_tree.hip, pure host code, checked against scikit-learn on the CPU by tests/test_hdbscan.py)
Labels agree with scikit-learn's up to the numbering of the clusters (the numbering follows the orientation of the spanning-tree
edges, which no two implementations share) and up to fp32-vs-fp64 ties on borderline points.  No CPU fallback: the points must be on
the GPU."""
import os
from typing import Optional

import numpy as np
import torch

from .. import _C

_SUPPORTED_C = (3, 8, 16)


def _pad_channels(x: torch.Tensor) -> torch.Tensor:
    C = x.shape[1]
    for c in _SUPPORTED_C:
        if C <= c:
            if C == c:
                return x.contiguous()
            out = torch.zeros(x.shape[0], c, dtype=torch.float32, device=x.device)
            out[:, :C] = x                       # zero columns do not change any distance
            return out
    raise _C.HipExtensionError(f"hdbscan: at most {_SUPPORTED_C[-1]} feature channels are built (got {C})")


def spatial_order(x: torch.Tensor) -> torch.Tensor:
    """Permutation (int64 [M]) that orders the rows of x [M, C] along a Morton curve over their first three principal axes: rows that
    are close in feature space end up close in memory, which is what lets the kernels skip far 256-row tiles by their bounding boxes.
    Only the ORDER is used -- any permutation gives the same (exact) result."""
    M, C = x.shape
    if M < 3 or not x.is_cuda:
        return torch.arange(M, dtype=torch.int64, device=x.device)
    from . import misc

    finite = torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0)
    p3 = _C.project3(finite, misc.pca_axes(finite, 3)) if C > 3 else finite.contiguous()
    if C < 3:      # fewer than three feature channels: the Morton kernel reads x, y, z -- missing axes are constant zero
        p3 = torch.cat([p3, p3.new_zeros(M, 3 - C)], 1)
    center = p3.mean(0)
    spread = float(p3.std(0).max())
    cell = max(6.0 * spread, 1e-30) / 1024.0
    codes = _C.knn_morton_codes(p3.contiguous(), center.tolist(), 1.0 / cell)
    return torch.sort(codes.long(), stable=True).indices


def _segment_min(values: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """Minimum of every run of `counts` consecutive entries (float32 / float64).  (scatter_reduce(amin) serialises its atomics when
    a million points share five components: 6.7 s of glue at 1.35 M points.)"""
    return torch.segment_reduce(values, "min", lengths=counts, unsafe=True)


LAST_STATS = {}   # reports: Boruvka rounds and component counts of the last mutual_reachability_mst call
COMPONENT_BOUND = os.environ.get("IGGT_HDB_COMPONENT_BOUND", "1") != "0"   # A/B switch of the per-component pruning (csrc/hdbscan.hip)


def mutual_reachability_mst(x: torch.Tensor, min_samples: int, _kernels=None):
    """x fp32 [M, C] on the GPU -> (eu, ev int64 [M - 1], ew fp32 [M - 1]) spanning-tree edges (original indices, weights =
    mutual-reachability distances) and the core distances fp32 [M].
    `_kernels`: (core_dist, nearest_foreign) stand-ins with the signatures of the two _C entry points -- ONLY for the CPU test of
    the round logic below (tests/test_hdbscan.py injects torch re-statements); the product never passes it."""
    core_dist, nearest_foreign = _kernels or (_C.hdbscan_core_dist, _C.hdbscan_nearest_foreign)
    if not x.is_cuda and _kernels is None:
        raise _C.HipExtensionError("hdbscan runs on the GPU (no CPU fallback)")
    x = _pad_channels(x.detach().float())
    M = x.shape[0]
    if min_samples > M:
        raise ValueError(f"min_samples ({min_samples}) must be at most the number of samples ({M})")
    dev = x.device
    perm = spatial_order(x)                                        # everything below works in the spatially sorted numbering
    x = x[perm].contiguous()
    core = core_dist(x, int(min_samples), _C.hdbscan_tile_boxes(x))
    core2 = core * core
    comp = torch.arange(M, dtype=torch.int64, device=dev)          # component id of every point = smallest member index
    big = float(2 ** 62)
    eu, ev, ew = [], [], []
    ncomp = M
    LAST_STATS.clear()
    LAST_STATS.update(points=M, components_per_round=[M])
    while ncomp > 1:
        comps, order = torch.sort(comp, stable=True)               # positions sorted by component, spatial order inside
        xs, c2s = x[order].contiguous(), core2[order].contiguous()
        ntile = (M + 255) // 256
        padded = torch.cat([comps, comps[-1:].expand(ntile * 256 - M)]).view(ntile, 256)
        extra = {} if _kernels is not None else {"component_bound": COMPONENT_BOUND}
        w2, bp = nearest_foreign(xs, c2s, comps.int().contiguous(), order.int().contiguous(),
                                 padded.amin(1).int().contiguous(), padded.amax(1).int().contiguous(), _C.hdbscan_tile_boxes(xs),
                                 **extra)
        # (with the component bound a point that cannot hold its component's cheapest edge reports (inf, -1): bp = -1 indexes the
        #  last position below, harmlessly -- only entries with w2 == the component's finite minimum are looked at)
        oi, oj = order, order[bp.long()]
        lo, hi = torch.minimum(oi, oj), torch.maximum(oi, oj)
        # cheapest outgoing edge of every component under the total order (weight, lo, hi): components are runs of `comps`
        roots, counts = torch.unique_consecutive(comps, return_counts=True)
        wmin = _segment_min(w2, counts)
        key = torch.where(w2 == torch.repeat_interleave(wmin, counts), (lo * M + hi).double(), torch.full_like(w2, big, dtype=torch.float64))
        ekey = _segment_min(key, counts).long()                    # lo * M + hi < 2^53: exact in float64
        elo, ehi = ekey // M, ekey % M
        # two components may pick the same edge: keep it once -- first occurrence per key, keys ascending (what numpy's
        # unique(return_index=True) returned when this step still copied the keys to the host): stable device sort + run heads
        skey, sidx = torch.sort(ekey, stable=True)
        head = torch.ones_like(skey, dtype=torch.bool)
        head[1:] = skey[1:] != skey[:-1]
        first = sidx[head]
        eu.append(elo[first])
        ev.append(ehi[first])
        ew.append(torch.sqrt(wmin[first]))
        # hook every component onto the component at the other end of its edge; a mutual pair keeps the smaller id as root
        ca, cb = comp[elo], comp[ehi]
        other = torch.where(ca == roots, cb, ca)
        parent = torch.arange(M, dtype=torch.int64, device=dev)
        parent[roots] = other
        mutual = (parent[parent[roots]] == roots) & (roots < other)
        parent[roots[mutual]] = roots[mutual]
        while True:                                                # pointer jumping
            nxt = parent[parent]
            if torch.equal(nxt, parent):
                break
            parent = nxt
        comp = parent[comp]
        n_new = int(torch.unique(comp).numel())
        if n_new >= ncomp:
            raise _C.HipExtensionError("hdbscan: a Boruvka round merged nothing (non-finite features?)")
        ncomp = n_new
        LAST_STATS["components_per_round"].append(n_new)
    eu, ev, ew = torch.cat(eu), torch.cat(ev), torch.cat(ew)
    if eu.numel() != M - 1:
        raise _C.HipExtensionError(f"hdbscan: spanning tree has {eu.numel()} edges for {M} points")
    core_out = torch.empty_like(core)
    core_out[perm] = core
    return perm[eu], perm[ev], ew, core_out


def hdbscan_labels(x: torch.Tensor, min_cluster_size: int, min_samples: Optional[int] = None,
                   cluster_selection_epsilon: float = 0.0, allow_single_cluster: bool = False, _kernels=None) -> np.ndarray:
    """x fp32 [M, C] on the GPU -> int32 labels [M] (numpy; -1 = noise), the estimator call of misc.py:123-129."""
    M = x.shape[0]
    if min_cluster_size is None or int(min_cluster_size) < 2:
        raise ValueError("min_cluster_size must be an integer >= 2")
    k = int(min_cluster_size if min_samples is None else min_samples)
    if M == 1:
        return np.full(1, -1, dtype=np.int32)
    eu, ev, ew, _ = mutual_reachability_mst(x, k, _kernels)
    # The host walk starts with a stable sort of the edges by weight -- 0.1 s of indirect comparisons on one core at 1.35 M edges.
    # Handing it the edges already stably sorted (device radix sort) leaves its result unchanged (equal weights keep their order
    # under both sorts) and turns its own sort into a linear pass.
    srt = torch.sort(ew, stable=True).indices
    eu, ev, ew = eu[srt], ev[srt], ew[srt]
    return _C.hdbscan_labels_from_mst(eu.cpu().numpy(), ev.cpu().numpy(), ew.cpu().numpy(), M, int(min_cluster_size),
                                      float(cluster_selection_epsilon or 0.0), allow_single_cluster)
