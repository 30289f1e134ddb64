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


def hdb_stats():   # only in the IGGT_HIP_LIB=probes/lib_alt/hdb_stats.so build
    import ctypes

    lib = _C.load()
    if not hasattr(lib, "iggt_hdb_stats"):
        return None
    buf = (ctypes.c_ulonglong * 8)()
    lib.iggt_hdb_stats(buf, 1)
    return list(buf)


def timed_nf(*a):
    hdb_stats()
    dt, out = sync_time(lambda: orig_nf(*a, component_bound=hd.COMPONENT_BOUND))   # IGGT_HDB_COMPONENT_BOUND=0: off
    st = hdb_stats()
    if st is not None:
        import ctypes
        import numpy as np

        nb = ((a[0].shape[0] + 511) // 512) * max(1, min(32, ((a[0].shape[0] + 255) // 256) // 32))   # x nsplit (_C.py)
        buf = (ctypes.c_ulonglong * (4 * nb))()
        _C.load().iggt_hdb_block_stats(buf, nb)
        blk = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 4).astype(np.float64)
        t0 = blk[:, 0].min()
        dur = (blk[:, 1] - blk[:, 0]) / 100.0            # us (100 MHz)
        start = (blk[:, 0] - t0) / 100.0
        order = np.argsort(-dur)[:5]
        st = st + [f"block durations us: mean {dur.mean():.0f} median {np.median(dur):.0f} max {dur.max():.0f}; last start at {start.max():.0f} us, "
                   f"last end at {((blk[:, 1] - t0) / 100.0).max():.0f} us; slowest blocks (id, us, steps, tiles): "
                   + ", ".join(f"({i}, {dur[i]:.0f}, {int(blk[i, 2])}, {int(blk[i, 3])})" for i in order)
                   + f"; us per step of blocks with no tile: {np.mean(dur[blk[:, 3] == 0] / np.maximum(blk[blk[:, 3] == 0, 2], 1)) if (blk[:, 3] == 0).any() else float('nan'):.2f}"
                   + f"; corr(duration, tiles) {np.corrcoef(dur, blk[:, 3])[0, 1]:.2f}"]
    comps = a[2]
    tl, th = a[4], a[5]
    single_tiles = int((tl == th).sum())
    pruned = int(torch.isinf(out[0]).sum())
    rounds.append((dt, int(torch.unique(comps).numel()), st, single_tiles, int(tl.numel()), pruned))
    return out


dt, _ = sync_time(lambda: hd.mutual_reachability_mst(x, 100, _kernels=(timed_cd, timed_nf)))
print(f"mutual_reachability_mst: {dt:.3f} s = core pass {sum(cores):.3f} s + {len(rounds)} Boruvka rounds, kernels "
      f"{sum(r[0] for r in rounds):.3f} s + glue")
for i, (dt, nc, st, stl, ntl, pruned) in enumerate(rounds):
    print(f"  round {i:2d}: {nc:8d} components, kernel {dt:.3f} s; single-component tiles {stl} of {ntl}; points reporting no edge {pruned}")
    if st is not None:
        print(f"            steps {st[0]}, not skipped as own component {st[1]}, tiles processed by single-component workgroups "
              f"{st[2]}, by mixed ones {st[3]}  (a processed tile = 256 x 512 pairs)")
        if len(st) > 8:
            print("            " + st[8])
