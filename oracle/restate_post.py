"""CPU restatement of the demo's post-processing step (reference iggt/utils/misc.py:24-78, 81-170, 272-331).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (iggt_official_amd/utils/misc.py) never does.

Third-party pieces the reference calls and that are NOT installed here (requirements.txt:280-282 pins torch-geometric 2.6.1,
torch_cluster 1.6.3, torch_scatter 2.1.2; hdbscan 0.8.40 at line 84) are restated from their published semantics:
  * torch_cluster.knn_graph(x, k, batch, loop=False): for every node the k nearest OTHER nodes of the same batch in
    Euclidean distance, as edges (source = neighbour, target = centre).  (The library searches k+1 and drops the self
    edge; with exact duplicates of a point its choice among equal distances is unspecified -- so is ours.)
  * torch_scatter.scatter_mean(src, index, dim=0, dim_size): per-target sum / max(count, 1).
PARITY PINNING: the kNN restatement is pinned only against those published semantics ("parity unpinned" for the library call
itself); apply_pca_colormap and the label fill are pinned against the reference functions run on fixed inputs
(oracle/make_golden_post.py -> tests/golden/post_misc.pt)."""
import numpy as np
import torch


def knn_graph(x: torch.Tensor, k: int, batch=None, loop: bool = False, chunk: int = 2048) -> torch.Tensor:
    """torch_cluster.knn_graph semantics on the CPU by brute force: edge_index int64 [2,E] (row 0 neighbour, row 1 centre)."""
    x = x.double()
    M = x.shape[0]
    if batch is None:
        batch = torch.zeros(M, dtype=torch.long)
    src, dst = [], []
    for s in range(0, M, chunk):
        e = min(s + chunk, M)
        d = torch.cdist(x[s:e], x, compute_mode="donot_use_mm_for_euclid_dist")
        d[batch[s:e, None] != batch[None, :]] = float("inf")
        if not loop:
            d[torch.arange(e - s), torch.arange(s, e)] = float("inf")
        kk = min(k, M - (0 if loop else 1))
        val, idx = torch.topk(d, kk, dim=1, largest=False)
        ok = torch.isfinite(val)
        centre = torch.arange(s, e)[:, None].expand_as(idx)
        src.append(idx[ok])
        dst.append(centre[ok])
    return torch.stack([torch.cat(src), torch.cat(dst)])


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size=None) -> torch.Tensor:
    assert dim == 0
    n = int(dim_size if dim_size is not None else index.max() + 1)
    out = torch.zeros((n,) + src.shape[1:], dtype=src.dtype)
    out.index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
    return out / cnt.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))


def knn_avg_features(points: torch.Tensor, features: torch.Tensor, k: int) -> torch.Tensor:
    """misc.py:24-78: (N,H,W,3), (N,H,W,F) -> (N,H,W,F), all views one batch."""
    N, H, W, F = features.shape
    p = points.reshape(-1, 3).float()
    f = features.reshape(-1, F).float()
    s, c = knn_graph(p, k)
    return scatter_mean(f[s], c, 0, N * H * W).view(N, H, W, F)


def knn_index_sets(points: torch.Tensor, k: int):
    """Sorted neighbour indices per point, int64 [M,k] (for bit-exact comparison of the neighbour SETS)."""
    p = points.reshape(-1, 3)
    s, c = knn_graph(p, k)
    M = p.shape[0]
    kk = s.numel() // M
    assert torch.equal(c, torch.arange(M).repeat_interleave(kk))
    return torch.sort(s.view(M, kk), dim=1).values


def pca_colormap(image: torch.Tensor, sign_like=None) -> torch.Tensor:
    """misc.py:272-331 with the randomised torch.pca_lowrank replaced by the exact SVD of the centred pixels (the reference's
    q = C makes its result the same axes up to sign).  sign_like: optional [C,3] axes whose orientation to adopt."""
    n, h, w, c = image.shape
    flat = image.reshape(-1, c).float()
    cen = (flat - flat.mean(0, keepdim=True)).double()
    _, _, vh = torch.linalg.svd(cen, full_matrices=False)
    v = vh.T[:, :3]
    if sign_like is not None:
        v = v * torch.sign((v * sign_like.double()).sum(0))
    else:
        big = v.abs().argmax(0)
        v = v * torch.sign(v[big, torch.arange(3)])
    col = flat @ v.float()
    for i in range(3):
        ch = col[:, i]
        lo, hi = torch.quantile(ch, 0.02), torch.quantile(ch, 0.98)
        col[:, i] = (ch - lo) / (hi - lo) if hi > lo else 0.5
    return col.clamp(0, 1).view(n, h, w, 3)


def fill_noise_labels(pixels: np.ndarray, labels: np.ndarray) -> np.ndarray:
    """misc.py:128-144: noise pixels (-1) take the label of the nearest labelled pixel in feature space."""
    labels = labels.copy()
    bad = labels == -1
    if bad.sum() == 0:
        return labels
    if bad.sum() == len(bad):
        return np.zeros_like(labels)
    good_px, good_lab = torch.from_numpy(pixels[~bad]).double(), labels[~bad]
    d = torch.cdist(torch.from_numpy(pixels[bad]).double(), good_px, compute_mode="donot_use_mm_for_euclid_dist")
    labels[bad] = good_lab[d.argmin(1).numpy()]
    return labels
