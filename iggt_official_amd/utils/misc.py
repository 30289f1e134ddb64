"""Post-processing of the part-feature maps behind the forward path, on the GPU (reference iggt/utils/misc.py, used by
demo.py:365-400: F.normalize -> apply_pca_colormap -> knn_avg_features_pyg -> apply_pca_colormap -> cluster_features_to_masks_mv).

Same names, argument meaning and return shapes as the reference functions; the arithmetic runs in csrc/postprocess.hip:

* `knn_avg_features_pyg` (misc.py:24-78): the reference builds torch_cluster's brute-force kNN graph over all S*H*W predicted
  points as one batch and averages the neighbours' features with torch_scatter.  Here: Morton sort + exact tile-pruned search
  + neighbour mean (no graph library needed, no O(M^2) pass).
* `apply_pca_colormap` (misc.py:272-331): the reference's torch.pca_lowrank is a RANDOMISED range finder (q = C columns, so it
  spans the full space and returns the exact principal axes up to sign -- the sign depends on the random draw).  Here the
  covariance is accumulated in two passes on the GPU and diagonalised in fp64; the sign of every axis is fixed (largest-
  magnitude component positive), so the colours are deterministic.  A flipped axis maps a channel c to 1 - c (the 2 % / 98 %
  stretch is symmetric); tests compare up to that flip.
* `cluster_features_to_masks_mv` (misc.py:81-170): HDBSCAN on the GPU (utils/hdbscan.py: exact core distances and the
  mutual-reachability spanning tree by Boruvka rounds as brute-force HIP kernels, csrc/hdbscan.hip; dendrogram / condensed tree /
  excess-of-mass / epsilon on the host, csrc/hdbscan_tree.hip) -- the reference calls cuml's, hdbscan's or scikit-learn's
  estimator (misc.py:19-22,123-129); labels agree with scikit-learn's up to the cluster numbering.  The nearest-labelled-pixel
  fill of the noise pixels and the colouring run on the GPU as well.
"""
from typing import Tuple, Union

import numpy as np
import torch

from .. import _C

__all__ = ["knn_avg_features_pyg", "knn_indices", "apply_pca_colormap", "cluster_features_to_masks_mv"]


def _as_device_f32(x, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(device=device, dtype=torch.float32)


def _device(device):
    dev = torch.device("cuda" if device in (None, "cpu") else device)   # the reference's default 'cpu' has no meaning here
    if dev.type != "cuda":
        raise _C.HipExtensionError("the post-processing kernels need a ROCm device (no CPU fallback)")
    return dev


def knn_indices(points: torch.Tensor, k: int, return_sq_dist: bool = False):
    """Exact k nearest neighbours (the point itself excluded) of every row of points fp32 [M,3] among all rows:
    int32 [M,k], ascending distance, -1 where fewer than k other points exist.  Same edge set as
    torch_cluster.knn_graph(points, k, batch=zeros, loop=False) (misc.py:61-65) apart from ties at the k-th distance."""
    assert points.dim() == 2 and points.shape[1] == 3
    pts = points.contiguous().float()
    # the grid only orders the points: centre / scale from robust-enough statistics (far outliers are clamped to the border)
    finite = torch.nan_to_num(pts, nan=0.0, posinf=0.0, neginf=0.0)
    center = finite.mean(0)
    spread = float(finite.std(0).max()) if pts.shape[0] > 1 else 1.0
    cell = max(6.0 * spread, 1e-30) / 1024.0
    codes = _C.knn_morton_codes(pts, center.tolist(), 1.0 / cell)
    order = torch.sort(codes.long(), stable=True).indices      # rocPRIM radix sort under torch.sort
    return _C.knn_search(pts, order, k, want_dist=return_sq_dist)


def knn_avg_features_pyg(points_batch, features_batch, k, device="cuda"):
    """points (N,H,W,3), features (N,H,W,F), k -> smoothed features (N,H,W,F): every pixel gets the mean feature of its k
    nearest pixels in 3-D over ALL views (reference misc.py:24-78)."""
    dev = _device(device)
    pts = _as_device_f32(points_batch, dev)
    feat = _as_device_f32(features_batch, dev)
    N, H, W, F = feat.shape
    idx = knn_indices(pts.reshape(-1, 3), int(k))
    out = _C.knn_mean_features(feat.reshape(-1, F).contiguous(), idx)
    return out.view(N, H, W, F)


def _quantile(x: torch.Tensor, q: float) -> torch.Tensor:
    """torch.quantile(x, q) (linear interpolation) without its 16 M element limit."""
    if x.numel() <= (1 << 24):
        return torch.quantile(x, q)
    s = torch.sort(x).values
    pos = q * (s.numel() - 1)
    lo = int(np.floor(pos))
    hi = min(lo + 1, s.numel() - 1)
    return torch.lerp(s[lo], s[hi], pos - lo)


def pca_axes(flat: torch.Tensor, n: int = 3) -> torch.Tensor:
    """The first n principal axes [C,n] (fp32, unit columns, largest-magnitude component positive) of the rows of flat [M,C]."""
    M, C = flat.shape
    s1, _ = _C.moments(flat, torch.zeros(C, device=flat.device))
    mean = (s1 / M).float()
    r1, g = _C.moments(flat, mean)                      # second pass around the mean: no cancellation
    cov = (g - torch.outer(r1, r1) / M) / max(M - 1, 1)
    w, v = torch.linalg.eigh(cov.cpu())                 # C x C, fp64
    v = v[:, torch.argsort(w, descending=True)[:n]]
    big = v.abs().argmax(0)
    v = v * torch.sign(v[big, torch.arange(v.shape[1])])
    return v.float().to(flat.device).contiguous()


def apply_pca_colormap(image: torch.Tensor) -> torch.Tensor:
    """(N,H,W,C) feature images -> (N,H,W,3) colours: one PCA over the pixels of all views, projection on the first three
    axes, per-channel 2 % .. 98 % stretch, clamp to [0,1] (reference misc.py:272-331)."""
    dev = _device(image.device if isinstance(image, torch.Tensor) and image.is_cuda else None)
    img = _as_device_f32(image, dev)
    n, h, w, c = img.shape
    flat = img.reshape(-1, c).contiguous()
    v = pca_axes(flat, 3)
    col = _C.project3(flat, v)
    lohi = torch.stack([_quantile(col[:, i], q) for q in (0.02, 0.98) for i in range(3)])
    _C.stretch3(col, lohi.float().contiguous())
    return col.view(n, h, w, 3)


def _hdbscan(pix: torch.Tensor, eps, min_samples, min_cluster_size) -> np.ndarray:
    """The estimator call of the reference (misc.py:123-129) on the GPU: iggt_official_amd/utils/hdbscan.py."""
    from .hdbscan import hdbscan_labels

    return hdbscan_labels(pix, min_cluster_size, min_samples, float(eps) if eps is not None else 0.0, allow_single_cluster=False)


FILL_TILED_MIN = 1e11   # noise x labelled pairs; below it the brute-force kernel (~1e12 pairs / s) beats sorting the pixels first


def fill_noise_labels(pixels: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """labels int [M] with -1 = noise -> every noise pixel takes the label of its nearest labelled pixel in feature space
    (all zeros when nothing is labelled), reference misc.py:128-144."""
    labels = labels.to(device=pixels.device, dtype=torch.int32).clone()
    bad = labels < 0
    nbad = int(bad.sum())
    if nbad == 0:
        return labels
    if nbad == labels.numel():
        return torch.zeros_like(labels)
    if float(nbad) * float(labels.numel() - nbad) < FILL_TILED_MIN or not torch.isfinite(pixels).all():
        labels[bad] = _C.nn1_label(pixels[bad].contiguous(), pixels[~bad].contiguous(), labels[~bad].contiguous())
        return labels
    # Large inputs: the same exact result from a local search.  All pixels are ordered along one Morton curve over their first
    # three principal axes (the order the HDBSCAN kernels use); noise pixels and labelled pixels keep that order, the kernel
    # skips sample tiles whose bounding box is farther than the best distance so far.  Ties go to the smallest pixel index,
    # like the first minimum of the brute-force kernel over pixels[~bad].
    from . import hdbscan as _hd

    perm = _hd.spatial_order(pixels)
    badp = bad[perm]
    q_order, r_order = perm[badp], perm[~badp]
    got = _C.nn1_label_tiled(pixels[q_order].contiguous(), pixels[r_order].contiguous(), r_order.int().contiguous(),
                             labels[r_order].contiguous())
    labels[q_order] = got
    return labels


def _jet_colors(n_colors: int) -> np.ndarray:
    import matplotlib.pyplot as plt

    cmap = plt.colormaps.get_cmap("jet")
    if n_colors > 1:
        return np.array([cmap(j / (n_colors - 1))[:3] for j in range(n_colors)], dtype=np.float64)
    return np.array([cmap(0.5)[:3]], dtype=np.float64)


def cluster_features_to_masks_mv(feature_map: Union[torch.Tensor, np.ndarray], apply_colormap: bool = False,
                                 clusterer=None, **kwargs) -> Union[np.ndarray, Tuple[np.ndarray, np.ndarray]]:
    """(N,H,W,C) features -> (N,H,W) integer masks clustered over all views together [, (N,H,W,3) uint8 colours]
    (reference misc.py:81-170).  kwargs: eps, min_samples, min_cluster_size.  HDBSCAN itself runs on the GPU (utils/hdbscan.py:
    exact core distances and mutual-reachability spanning tree as HIP kernels, the tree walk on the host);
    `clusterer(pixels ndarray [M,C]) -> labels [M]` replaces it (other clustering back-ends, tests)."""
    if not (isinstance(feature_map, (torch.Tensor, np.ndarray)) and feature_map.ndim == 4):
        raise ValueError("feature_map must be a 4-D tensor or array of shape (N, H, W, C)")
    n, h, w, c = feature_map.shape
    dev = _device(feature_map.device if isinstance(feature_map, torch.Tensor) and feature_map.is_cuda else None)
    pix = _as_device_f32(feature_map, dev).reshape(-1, c).contiguous()
    raw = clusterer(pix.cpu().numpy()) if clusterer is not None else _hdbscan(pix, kwargs.get("eps"), kwargs.get("min_samples"),
                                                                               kwargs.get("min_cluster_size"))
    labels = fill_noise_labels(pix, torch.as_tensor(np.asarray(raw).astype(np.int64)))
    masks = labels.view(n, h, w)
    if not apply_colormap:
        return masks.cpu().numpy().astype(np.int64)
    uniq = torch.unique(labels)                     # sorted; no -1 left after the fill
    colors = torch.from_numpy((_jet_colors(len(uniq)) * 255).astype(np.uint8)).to(dev)
    rank = torch.searchsorted(uniq, labels)
    colored = colors[rank].view(n, h, w, 3)
    return masks.cpu().numpy().astype(np.int64), colored.cpu().numpy()
