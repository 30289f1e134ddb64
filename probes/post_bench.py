#!/usr/bin/env python3
"""Timing of the post-processing step (iggt_official_amd/utils/misc.py) at demo scale: kNN feature averaging, PCA colour map,
noise-label fill.  Synthetic multi-view surfaces (same generator as oracle/make_golden_post.py)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C  # noqa: E402
from iggt_official_amd.utils import misc  # noqa: E402
from oracle.make_golden_post import scene  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--h", type=int, default=336)
    ap.add_argument("--w", type=int, default=504)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--brute-rows", type=int, default=20000, help="rows of the brute-force cross-check (0: skip)")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of a host KD-tree query for comparison (0: skip)")
    ap.add_argument("--hdbscan", action="store_true", help="also time the clustering step (GPU HDBSCAN, demo.py:78-83 parameters)")
    a = ap.parse_args()
    pts, feats = scene(a.views, a.h, a.w, 8, seed=3)
    pts, feats = pts.cuda(), feats.cuda()
    M = pts.numel() // 3
    res = {"views": a.views, "h": a.h, "w": a.w, "points": M, "k": a.k}
    res["knn_avg_ms"], smooth = timed(lambda: misc.knn_avg_features_pyg(pts, feats, a.k))
    flat = pts.reshape(-1, 3).contiguous()
    res["knn_search_only_ms"], idx = timed(lambda: misc.knn_indices(flat, a.k))
    res["pca_colormap_ms"], _ = timed(lambda: misc.apply_pca_colormap(smooth))
    lab = ((smooth[..., 0] > 0).int() * 2 + (smooth[..., 1] > 0).int()).reshape(-1)
    lab[torch.rand(M, device="cuda") < 0.3] = -1
    px = smooth.reshape(-1, 8).contiguous()
    res["label_fill_ms"], _ = timed(lambda: misc.fill_noise_labels(px, lab), reps=1)
    if a.hdbscan:
        from iggt_official_amd.utils import hdbscan as hd

        torch.cuda.synchronize()
        t = time.perf_counter()
        eu, ev, ew, core = hd.mutual_reachability_mst(px, 100)
        torch.cuda.synchronize()
        res["hdbscan_core_and_spanning_tree_s"] = time.perf_counter() - t
        res["hdbscan_boruvka_components_per_round"] = hd.LAST_STATS.get("components_per_round")
        t = time.perf_counter()
        labels = _C.hdbscan_labels_from_mst(eu.cpu().numpy(), ev.cpu().numpy(), ew.cpu().numpy(), M, 500, 0.06, False)
        res["hdbscan_tree_walk_host_s"] = time.perf_counter() - t
        res["hdbscan_clusters"], res["hdbscan_noise"] = int(labels.max() + 1), int((labels < 0).sum())
        t = time.perf_counter()
        misc.cluster_features_to_masks_mv(smooth, apply_colormap=True, eps=0.06, min_samples=100, min_cluster_size=500)
        res["cluster_features_to_masks_mv_s"] = time.perf_counter() - t
    if a.brute_rows:
        rows = torch.randperm(M, device="cuda")[: a.brute_rows]
        t = time.perf_counter()
        d = torch.cdist(flat[rows].double(), flat.double())
        d[torch.arange(rows.numel(), device="cuda"), rows] = float("inf")
        ref = torch.sort(torch.topk(d, a.k, dim=1, largest=False).indices, 1).values
        torch.cuda.synchronize()
        res["brute_force_ms_per_1k_rows"] = (time.perf_counter() - t) * 1e3 / (a.brute_rows / 1000)
        got = torch.sort(idx[rows].long(), 1).values
        res["rows_checked"] = int(rows.numel())
        res["rows_with_identical_neighbour_sets"] = int((got == ref).all(1).sum())
    if a.cpu_rows:
        from sklearn.neighbors import NearestNeighbors

        host = flat.cpu().numpy()
        t = time.perf_counter()
        nn = NearestNeighbors(n_neighbors=a.k + 1, algorithm="kd_tree", n_jobs=-1).fit(host)
        res["cpu_kdtree_build_s"] = time.perf_counter() - t
        t = time.perf_counter()
        nn.kneighbors(host[: a.cpu_rows])
        res["cpu_kdtree_query_s_per_100k_rows"] = (time.perf_counter() - t) / (a.cpu_rows / 1e5)
        res["cpu_cores"] = os.cpu_count()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
