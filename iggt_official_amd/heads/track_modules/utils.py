"""Sampling / embedding helpers of the tracker -- reference iggt/heads/track_modules/utils.py.

`get_2d_sincos_pos_embed` builds a [1, D, H, W] table whose first D/2 channels depend on x only and the others on y
only; the tracker only ever samples it at the query points (base_track_predictor.py:152-154), so the product keeps
the two 1-D tables ([W, D/2] and [H, D/2], built in fp64 exactly like the reference's: utils.py:66-87) and the sampling
kernel `iggt_track_posemb_f32` interpolates those.  The 2-D table and the torch samplers of the reference are not
rebuilt: `sample_features4d` is `iggt_sample_points_nhwc_f32` on NHWC maps."""
import torch

from ... import _C

_TABLES = {}


def sincos_tables(embed_dim: int, grid_size, device):
    """-> (tabx [W, embed_dim / 2], taby [H, embed_dim / 2]) fp32 on `device`: [sin | cos](pos * 10000^(-j / (D/4)))."""
    H, W = (grid_size, grid_size) if isinstance(grid_size, int) else grid_size
    key = (embed_dim, H, W, str(device))
    if key not in _TABLES:
        assert embed_dim % 4 == 0
        q = embed_dim // 4
        omega = torch.arange(q, dtype=torch.double)
        omega /= q
        omega = 1.0 / 10000 ** omega                                         # utils.py:78-80
        tabs = []
        for n in (W, H):
            ang = torch.arange(n, dtype=torch.float).double()[:, None] * omega[None]     # utils.py:82-83 (float grid x fp64)
            tabs.append(torch.cat([ang.sin(), ang.cos()], 1).float().contiguous().to(device))
        _TABLES[key] = tuple(tabs)
    return _TABLES[key]


def sample_features4d(fmap_nhwc: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """fmap_nhwc [H, W, C] fp32 (one frame), coords [R, 2] pixel (x, y) -> [R, C]; bilinear, align_corners=True, border
    padding (utils.py:192-226 on the product's NHWC layout)."""
    if not fmap_nhwc.is_cuda:
        raise _C.HipExtensionError("track head runs on HIP kernels only (no CPU fallback)")
    return _C.sample_points_nhwc(fmap_nhwc.contiguous(), coords.float().contiguous())
