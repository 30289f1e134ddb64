"""Input-independent sinusoidal UV position maps for the DPT heads (host-side tables).

Same function as reference iggt/heads/utils.py:11-108 (`create_uv_grid`, `position_grid_to_embed`,
`make_sincos_pos_embed`) + `_apply_pos_embed` (iggt/heads/dpt_head.py:274-284), collapsed into one
closed form (verified bit-exact against the reference) and cached per
(C, h, w, W/H, device): the map depends on no activation, so it is built once in float64 on the host
(the reference's precision: omega in double, utils.py:48-58) and reused by every forward.
"""
import torch

_CACHE = {}


def uv_grid(w: int, h: int, aspect: float) -> torch.Tensor:
    """[h, w, 2] (u, v) grid spanning +-(aspect, 1)/sqrt(aspect^2+1) * (n-1)/n, float32 linspace."""
    diag = (aspect ** 2 + 1.0) ** 0.5
    sx, sy = aspect / diag, 1.0 / diag
    xs = torch.linspace(-sx * (w - 1) / w, sx * (w - 1) / w, steps=w, dtype=torch.float32)
    ys = torch.linspace(-sy * (h - 1) / h, sy * (h - 1) / h, steps=h, dtype=torch.float32)
    uu, vv = torch.meshgrid(xs, ys, indexing="xy")
    return torch.stack((uu, vv), dim=-1)


def sincos_embed(grid: torch.Tensor, C: int, omega_0: float = 100.0) -> torch.Tensor:
    """[h, w, 2] -> [h, w, C] = [sin(u w), cos(u w), sin(v w), cos(v w)], w_j = omega_0^(-j/(C/4))."""
    h, w, _ = grid.shape
    q = C // 4
    omega = torch.arange(q, dtype=torch.float64) / float(q)
    omega = 1.0 / omega_0 ** omega
    flat = grid.reshape(-1, 2)
    parts = []
    for d in range(2):
        ang = torch.einsum("m,d->md", flat[:, d], omega)  # float32 pos x float64 omega -> float64
        parts += [torch.sin(ang), torch.cos(ang)]
    return torch.cat(parts, dim=1).float().view(h, w, C)


def pos_embed_map(C: int, h: int, w: int, W: int, H: int, device, ratio: float = 0.1) -> torch.Tensor:
    """ratio * embed as an NHWC-flattened fp32 table [h*w, C] on `device`."""
    key = (C, h, w, W, H, str(device), ratio)
    if key not in _CACHE:
        emb = sincos_embed(uv_grid(w, h, W / H), C) * ratio
        _CACHE[key] = emb.reshape(h * w, C).contiguous().to(device)
    return _CACHE[key]


def pos_embed_xy(C: int, h: int, w: int, W: int, H: int, device, ratio: float = 0.1):
    """The map is separable: channels [0, C/2) depend on x only, [C/2, C) on y only.
    Returns (xpart [1, C/2, 1, w], ypart [1, C/2, h, 1]) fp32, already scaled by `ratio`."""
    key = ("xy", C, h, w, W, H, str(device), ratio)
    if key not in _CACHE:
        g = uv_grid(w, h, W / H)
        q = C // 4
        omega = 1.0 / 100.0 ** (torch.arange(q, dtype=torch.float64) / float(q))
        ax = torch.einsum("m,d->md", g[0, :, 0], omega)  # [w, q] float64
        ay = torch.einsum("m,d->md", g[:, 0, 1], omega)  # [h, q]
        xp = (torch.cat([torch.sin(ax), torch.cos(ax)], 1).float() * ratio).t().reshape(1, 2 * q, 1, w)
        yp = (torch.cat([torch.sin(ay), torch.cos(ay)], 1).float() * ratio).t().reshape(1, 2 * q, h, 1)
        _CACHE[key] = (xp.contiguous().to(device), yp.contiguous().to(device))
    return _CACHE[key]


def pos_embed_rows(C: int, h: int, w: int, W: int, H: int, device, ratio: float = 0.1):
    """Separable form for the NHWC resize kernel: (xrows [w, C/2], yrows [h, C/2]) fp32 contiguous."""
    key = ("rows", C, h, w, W, H, str(device), ratio)
    if key not in _CACHE:
        xp, yp = pos_embed_xy(C, h, w, W, H, device, ratio)
        rows = (xp[0, :, 0, :].t().contiguous(), yp[0, :, :, 0].t().contiguous())
        # the two transposes above are kernels on the CURRENT stream; the cache is read from other streams too (the depth and
        # point heads run side by side, models/vggt.py): an entry must be complete when it becomes visible.  (The other entries
        # are host tensors copied synchronously.)  Never reached under graph capture: the eager warm-up run fills the cache.
        if xp.is_cuda and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(xp.device).synchronize()
        _CACHE[key] = rows
    return _CACHE[key]
