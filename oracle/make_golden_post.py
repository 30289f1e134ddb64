#!/usr/bin/env python3
"""Golden vectors for the demo post-processing step, produced by the REFERENCE functions (build container only).

TEST INFRASTRUCTURE.  Usage: python oracle/make_golden_post.py  ->  tests/golden/post_misc.pt
Imports /root/reference/iggt/utils/misc.py through oracle/ref_shim.py.  That module imports packages that are not installed
here (cv2, jaxtyping, torch_geometric, torch_scatter, hdbscan / cuml) and iggt.utils.vo_eval; they are replaced by stubs
first: inert ones for what these three functions never touch, and for what they do touch
  torch_geometric.nn.knn_graph / torch_scatter.scatter_mean -> oracle/restate_post.py (published semantics, brute force),
  hdbscan.HDBSCAN -> a deterministic stand-in (labels supplied by this script), because the fixture pins the reference's
  label fill and colouring, not a clustering library.
So the fixture holds: apply_pca_colormap = the reference function itself (torch.manual_seed(0) in front of its randomised
pca_lowrank); knn_avg_features_pyg = the reference's glue around the restated graph call; cluster_features_to_masks_mv =
the reference's fill + colour code around given cluster labels."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, restate_post  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "post_misc.pt")


class _Anything:
    def __getitem__(self, item):
        return self

    def __call__(self, *a, **k):
        return self


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def scene(N, H, W, F, seed):
    """A synthetic multi-view point map: a few smooth surfaces seen from N shifted viewpoints, plus far outliers; features =
    noisy unit vectors that vary smoothly with the surface position."""
    g = torch.Generator().manual_seed(seed)
    v, u = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    pts, feats = [], []
    basis = torch.randn(3, F, generator=g)
    for n in range(N):
        z = 2.0 + 0.3 * torch.sin(3 * u + n) + 0.2 * v * v + (u > 0.2).float() * 0.8
        p = torch.stack([u * z + 0.1 * n, v * z, z], -1) + 0.002 * torch.randn(H, W, 3, generator=g)
        far = torch.rand(H, W, generator=g) < 0.01
        p[far] *= 40.0
        f = torch.tanh(p @ basis) + 0.05 * torch.randn(H, W, F, generator=g)
        pts.append(p)
        feats.append(torch.nn.functional.normalize(f, dim=-1))
    return torch.stack(pts), torch.stack(feats)


def main():
    ref_shim.install()
    import iggt.utils  # noqa: F401
    _stub("cv2")
    _stub("jaxtyping", Float=_Anything())
    _stub("torch_geometric")
    _stub("torch_geometric.nn", knn_graph=restate_post.knn_graph)
    _stub("torch_scatter", scatter_mean=restate_post.scatter_mean)
    _stub("iggt.utils.vo_eval", save_trajectory_tum_format=_Anything())
    given = {}

    class HDBSCAN:                                   # stands in for hdbscan.HDBSCAN: returns the labels this script planted
        def __init__(self, **kw):
            given["kwargs"] = kw

        def fit(self, x):
            self.labels_ = given["labels"].copy()
            return self

    _stub("hdbscan", HDBSCAN=HDBSCAN)
    sys.modules.pop("cuml", None)
    import iggt.utils.misc as ref_misc

    for f in (ref_misc.knn_avg_features_pyg, ref_misc.apply_pca_colormap, ref_misc.cluster_features_to_masks_mv):
        ref_shim.assert_reference(f)

    N, H, W, F, k = 3, 24, 36, 8, 20
    pts, feats = scene(N, H, W, F, seed=3)
    smooth = ref_misc.knn_avg_features_pyg(pts, feats, k, device="cpu")
    torch.manual_seed(0)
    pca_raw = ref_misc.apply_pca_colormap(feats.clone())
    torch.manual_seed(0)
    pca_smooth = ref_misc.apply_pca_colormap(smooth.clone())

    # clustering: planted labels = sign pattern of the first two feature channels, 30 % of the pixels marked as noise
    flat = smooth.reshape(-1, F).numpy()
    lab = (flat[:, 0] > 0).astype(np.int64) * 2 + (flat[:, 1] > 0).astype(np.int64)
    rng = np.random.default_rng(5)
    lab[rng.random(lab.shape[0]) < 0.3] = -1
    given["labels"] = lab
    masks, colored = ref_misc.cluster_features_to_masks_mv(smooth, apply_colormap=True, eps=0.06, min_samples=100,
                                                           min_cluster_size=500)
    torch.save(dict(points=pts, features=feats, k=k, knn_avg=smooth, pca_raw=pca_raw, pca_smooth=pca_smooth,
                    planted_labels=torch.from_numpy(lab), masks=torch.from_numpy(np.asarray(masks)),
                    colored=torch.from_numpy(np.asarray(colored)), hdbscan_kwargs=given["kwargs"],
                    torch=torch.__version__, numpy=np.__version__), OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
