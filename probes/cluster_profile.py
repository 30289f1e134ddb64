#!/usr/bin/env python3
"""cProfile of cluster_features_to_masks_mv at the demo's size (8 x 336 x 504 pixels, 8 channels): where the host-side time of the
clustering step goes once the kernels are fast.  Usage: python probes/cluster_profile.py [views=8]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd.utils import misc  # noqa: E402
from oracle.make_golden_post import scene  # noqa: E402

views = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pts, feats = scene(views, 336, 504, 8, seed=3)
smooth = misc.knn_avg_features_pyg(pts.cuda(), feats.cuda(), 20)
kw = dict(apply_colormap=True, eps=0.06, min_samples=100, min_cluster_size=500)
misc.cluster_features_to_masks_mv(smooth, **kw)          # warm-up (library load, allocator)
torch.cuda.synchronize()
t = time.perf_counter()
misc.cluster_features_to_masks_mv(smooth, **kw)
print(f"cluster_features_to_masks_mv, second call: {time.perf_counter() - t:.3f} s")
os.environ["AMD_SERIALIZE_KERNEL"] = "0"
pr = cProfile.Profile()
pr.enable()
misc.cluster_features_to_masks_mv(smooth, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
