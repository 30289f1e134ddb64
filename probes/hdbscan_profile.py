#!/usr/bin/env python3
"""Where the GPU HDBSCAN spends its time: core-distance pass, and per Boruvka round the kernel, the component count and the
torch glue.  Usage: python probes/hdbscan_profile.py [views=8]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C  # noqa: E402
from iggt_official_amd.utils import hdbscan as hd, misc  # noqa: E402
from oracle.make_golden_post import scene  # noqa: E402

views = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pts, feats = scene(views, 336, 504, 8, seed=3)
smooth = misc.knn_avg_features_pyg(pts.cuda(), feats.cuda(), 20)
x = smooth.reshape(-1, 8).contiguous()
M = x.shape[0]


def sync_time(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t, out


dt, perm = sync_time(lambda: hd.spatial_order(x))
print(f"spatial order (PCA axes + Morton codes + sort): {dt:.3f} s")
xs = x[perm].contiguous()
boxes = _C.hdbscan_tile_boxes(xs)
for k in (100, 20, 5):
    dt, core = sync_time(lambda: _C.hdbscan_core_dist(xs, k, boxes))
    print(f"core distances k={k:3d}, spatially sorted: {dt:.3f} s for {M} points", flush=True)
dt, core = sync_time(lambda: _C.hdbscan_core_dist(x, 100))
print(f"core distances k=100, image order:        {dt:.3f} s", flush=True)
rounds, cores = [], []
orig_nf, orig_cd = _C.hdbscan_nearest_foreign, _C.hdbscan_core_dist


def timed_cd(*a):
    dt, out = sync_time(lambda: orig_cd(*a))
    cores.append(dt)
    return out


def timed_nf(*a):
    dt, out = sync_time(lambda: orig_nf(*a))
    rounds.append((dt, int(torch.unique(a[2]).numel())))
    return out


dt, _ = sync_time(lambda: hd.mutual_reachability_mst(x, 100, _kernels=(timed_cd, timed_nf)))
print(f"mutual_reachability_mst: {dt:.3f} s = core pass {sum(cores):.3f} s + {len(rounds)} Boruvka rounds, kernels "
      f"{sum(r[0] for r in rounds):.3f} s + glue")
for i, (dt, nc) in enumerate(rounds):
    print(f"  round {i:2d}: {nc:8d} components, kernel {dt:.3f} s")
