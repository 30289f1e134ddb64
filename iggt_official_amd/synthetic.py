"""Deterministic synthetic checkpoints and inputs for the IGGT state-dict schema.

No checkpoint is reachable (the official weights live on the HF hub, there is no network), so benchmarks, the smoke
test and the parity fixtures all run on seeded synthetic weights that are a pure function of
(tensor name, shape, seed, mode): the same function fills the REFERENCE model in the build container
(oracle/make_golden.py, through the re-export oracle/weights.py), the HIP model on the GPU box (tests/) and the model
bench.py times -- which is what lets bench.py check the outputs of the timed configuration against the reference
fixture tests/golden/full_s32_518_stress.pt.  Pure torch integer arithmetic, no dependency on anything else in this
package (and none on oracle/).

Modes
  "default": LayerScale gamma = 0.01 in aggregator/camera blocks and 1.0 in the DINOv2
             backbone, i.e. the reference's init scales (aggregator.py:63,153).
  "stress" : gamma ~ U(0.5, 1.5) everywhere, so that an error in any attention / MLP kernel
             reaches the outputs un-attenuated.
  "trained_like": "stress" plus the statistics in which a trained DINOv2-with-registers / VGGT-family checkpoint
             departs from bounded-uniform draws (round-3 review, item 1b): log-normal (sigma = 1) scales on every
             q_norm / k_norm / norm1 / norm2 weight (reference attention.py:43-44,54: learned affines), GAUSSIAN Linear /
             Conv weights with ~0.1 % of the input columns at 8x, and camera / register tokens at 30x the norm of a
             patch token (reference aggregator.py:123-124: attention-sink / high-norm tokens).  The draws go through
             Python-built fp32 tables (inverse normal CDF, exp) indexed by the integer hash, so they stay bit-identical
             between CPU and GPU -- no device transcendental is involved.
All other tensors are drawn so that activations stay O(1) through the network.
"""
import math
import statistics
import zlib

import torch


def _name_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) + 1000003 * seed) & 0x7FFFFFFF


def hash_uniform(n: int, key: int, device="cpu") -> torch.Tensor:
    """n reproducible U[0,1) fp32 samples from a counter-based integer hash.

    Only exact integer ops (every intermediate < 2**62, so no int64 wrap) and one exact
    int->float conversion: bit-identical on CPU and on the GPU, so the GPU box can synthesise the
    1.3 B parameters on-device in milliseconds instead of streaming a CPU RNG."""
    x = torch.arange(n, dtype=torch.int64, device=device)
    x = (x * 747796405 + key) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x45D9F3B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x45D9F3B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return (x >> 8).to(torch.float32) * (1.0 / 16777216.0)


_SQRT12 = 12.0 ** 0.5

# --- trained_like: table-driven Gaussian / log-normal draws (bit-identical on every device) --------------------------------
_NQ = 4096                 # inverse-CDF knots: z_k = Phi^-1((k + 0.5) / _NQ), |z| <= 3.66
_TABLES = {}


def _tables(device):
    key = str(device)
    if key not in _TABLES:
        nd = statistics.NormalDist()
        z = [nd.inv_cdf((k + 0.5) / _NQ) for k in range(_NQ)]
        zt = torch.tensor(z + [z[-1]], dtype=torch.float32)                    # one guard knot for the interpolation
        _TABLES[key] = zt.to(device)
    return _TABLES[key]


def hash_normal(n: int, key: int, device="cpu") -> torch.Tensor:
    """n reproducible ~N(0,1) fp32 samples: the hash's 24 bits pick a knot of the inverse normal CDF (upper 12 bits) and
    interpolate linearly to the next one (lower 12 bits); exact fp32 multiply-adds on table values only."""
    zt = _tables(device)
    u = hash_uniform(n, key, device) * float(_NQ)          # exact: k / 2**24 * 2**12
    idx = u.floor()
    frac = u - idx                                           # exact
    idx = idx.to(torch.int64)
    lo, hi = zt[idx], zt[idx + 1]
    return lo + frac * (hi - lo)


def hash_lognormal(n: int, key: int, device="cpu", sigma: float = 1.0) -> torch.Tensor:
    """n reproducible exp(sigma * N(0,1)) fp32 samples (exponent knots quantised to 1/16; range e^(+-3.7 sigma))."""
    tkey = (str(device), float(sigma))
    if tkey not in _TABLES:
        nd = statistics.NormalDist()
        _TABLES[tkey] = torch.tensor([math.exp(round(sigma * nd.inv_cdf((k + 0.5) / _NQ) * 16.0) / 16.0)
                                      for k in range(_NQ)], dtype=torch.float32).to(device)
    idx = (hash_uniform(n, key, device) * float(_NQ)).floor().to(torch.int64)
    return _TABLES[tkey][idx]


# "trained_like" takes its knobs from the mode string itself, so that a fixture's meta["mode"] names them all:
#   trained_like                      = the defaults below
#   trained_like(qk=0.5,tok=10)       = overrides
# Round 6 -- OUTLIER CHANNELS (what trained ViTs, DINOv2 included, are known for, as opposed to a heavy tail over ALL channels):
#   out=K, outmag=M   K channels of every norm1 / norm2 scale are multiplied by M (K = 1, M = 10: "one gamma = 10 among 1 023 of ~1");
#   outshare=1        the SAME K channels in every block (0: each LayerNorm draws its own);
#   massive=A         (with outshare=1) rows of every mlp.fc2 weight / bias that write those channels are multiplied by A: the
#                     residual stream then carries "massive activations" on them, which every later LayerNorm sees.
_TL_DEFAULTS = dict(qk=1.0, norm=1.0, tok=30.0, col=8.0, colfrac=1e-3, gauss=1.0, out=0.0, outmag=10.0, outshare=0.0, massive=1.0)


def trained_like_options(mode: str) -> dict:
    o = dict(_TL_DEFAULTS)
    if "(" in mode:
        body = mode[mode.index("(") + 1:mode.rindex(")")]
        for item in filter(None, (t.strip() for t in body.split(","))):
            k, v = item.split("=")
            if k not in o:
                raise ValueError(f"unknown trained_like option {k!r}")
            o[k] = float(v)
    return o


def outlier_channels(name: str, C: int, seed: int, o: dict, device="cpu") -> torch.Tensor:
    """The K outlier channel indices (int64 [K]) of the LayerNorm / fc2 tensor `name` (see _TL_DEFAULTS)."""
    K = int(o["out"])
    src = "#outlier-channels" if o["outshare"] else name.rsplit(".", 1)[0] + "#outlier-channels"
    return (hash_uniform(K, _name_seed(src, seed), device) * float(C)).floor().to(torch.int64)


def make_tensor(name: str, shape, seed: int, mode: str, device="cpu") -> torch.Tensor:
    """Synthetic value of state-dict entry `name`.  All draws are uniform (centred draws are scaled
    to the requested standard deviation)."""
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    n = 1
    for s_ in shape:
        n *= s_
    key = _name_seed(name, seed)

    def randn(std=1.0):  # zero-mean uniform with the given std
        return ((hash_uniform(n, key, device) - 0.5) * (_SQRT12 * std)).view(shape)

    def uniform(lo, hi):
        return (hash_uniform(n, key, device) * (hi - lo) + lo).view(shape)

    def one_plus(std):
        return ((hash_uniform(n, key, device) - 0.5) * (_SQRT12 * std) + 1.0).view(shape)

    heavy = mode.startswith("trained_like")
    if heavy:
        o = trained_like_options(mode)
        if leaf == "weight" and len(shape) == 1:
            parent = name.split(".")[-2]
            sig = o["qk"] if parent in ("q_norm", "k_norm") else o["norm"] if parent in ("norm1", "norm2") else 0.0
            if parent in ("norm1", "norm2") and o["out"] >= 1.0:
                w = hash_lognormal(n, key, device, sig).view(shape) if sig > 0.0 else one_plus(0.1)
                w[outlier_channels(name, n, seed, o, device)] *= o["outmag"]
                return w
            if sig > 0.0:
                return hash_lognormal(n, key, device, sig).view(shape)
        if (o["massive"] != 1.0 and o["outshare"] and o["out"] >= 1.0 and ".mlp.fc2." in name and ".blocks." in name.replace("_blocks.", ".blocks.")
                and shape[0] == 1024):
            base = make_tensor(name, shape, seed, mode.split("(")[0] + "(" + ",".join(
                f"{k}={v}" for k, v in o.items() if k != "massive") + ")", device)
            base[outlier_channels(name, shape[0], seed, o, device)] *= o["massive"]
            return base
        if leaf in ("camera_token", "register_token") and o["tok"] > 0.0:
            return (hash_normal(n, key, device) * o["tok"]).view(shape)
        if leaf == "weight" and len(shape) >= 2 and not name.endswith("updateformer.flow_head.weight") and o["gauss"]:
            conv_t = ".resize_layers." in name and len(shape) == 4 and _is_conv_transpose(name)
            if conv_t:
                fan = shape[0] * (4 if shape[2] == 4 and "part_adaptor" in name else 1)
            else:
                fan = 1
                for s_ in shape[1:]:
                    fan *= s_
            w = (hash_normal(n, key, device) * (fan ** -0.5)).view(shape)
            if o["col"] > 1.0:
                # ~0.1 % of the INPUT columns (dim 1 of [out, in, ...]; dim 0 of a ConvTranspose2d's [in, out, kh, kw])
                cdim = 0 if conv_t else 1
                pick = hash_uniform(shape[cdim], _name_seed(name + "#outlier", seed), device) < o["colfrac"]
                view = [1] * len(shape)
                view[cdim] = shape[cdim]
                w = w * torch.where(pick, o["col"], 1.0).to(torch.float32).view(view)
            return w
        mode = "stress"          # everything else: as in "stress"
    if leaf == "gamma":  # LayerScale
        if mode == "stress":
            return uniform(0.5, 1.5)
        return torch.full(shape, 1.0 if ".patch_embed.blocks." in name else 0.01, device=device)
    if leaf == "running_var":
        return uniform(0.5, 1.5)
    if leaf == "running_mean":
        return randn(0.1)
    if leaf in ("camera_token", "register_token", "cls_token", "register_tokens", "mask_token"):
        return randn(1.0 if mode == "stress" else 0.02)
    if leaf == "pos_embed":
        return randn(0.2)
    if leaf == "empty_pose_tokens":
        return randn(0.1)
    if leaf == "relative_position_bias_table":
        return randn(0.5)
    # track head (reference heads/track_modules/base_track_predictor.py:52, blocks.py:52 -- both torch.randn at init;
    # nn.MultiheadAttention keeps its input projection as in_proj_weight / in_proj_bias)
    if leaf in ("query_ref_token", "virual_tracks"):
        return randn(1.0)
    if leaf == "in_proj_bias":
        return randn(0.1)
    if leaf == "in_proj_weight":
        return randn(shape[1] ** -0.5)
    if name.endswith("updateformer.flow_head.weight"):
        # the reference initialises this layer with std 0.001 (blocks.py:97).  The tracker feeds coordinate differences
        # back through a sin / cos embedding of up to 969 rad per pixel (utils.py:107): with an O(1) flow head one
        # refinement iteration amplifies a perturbation ~40x and four iterations turn fp32 rounding noise into 0.3 pixels
        # in the REFERENCE itself; at std 0.005 an iteration moves a track by ~0.1-0.3 pixels and stays well-conditioned
        return randn(0.005)
    if leaf == "bias":
        # norm-layer bias or linear/conv bias: small but non-zero
        return randn(0.1)
    if leaf == "weight":
        if len(shape) == 1:  # LayerNorm / BatchNorm scale
            return one_plus(0.1)
        if ".resize_layers." in name and len(shape) == 4 and _is_conv_transpose(name):
            # ConvTranspose2d weight is [in, out, kh, kw]; each output pixel sums over
            # `in * ceil(k/stride)^2` taps.
            fan = shape[0] * (4 if shape[2] == 4 and "part_adaptor" in name else 1)
            return randn(fan ** -0.5)
        fan_in = 1
        for s_ in shape[1:]:
            fan_in *= s_
        return randn(fan_in ** -0.5)
    raise KeyError(f"no synthetic rule for {name} {shape}")


def _is_conv_transpose(name: str) -> bool:
    # DPTHead / PartHead(inherited): resize_layers.0 (k4 s4) and .1 (k2 s2) are ConvTranspose2d
    # (dpt_head.py:72-79).  SamProjector: resize_layers.0.0, .0.2 (k4 s2 p1) and .1.0 (k2 s2)
    # (adaptor.py:152-166).
    tail = name.split(".resize_layers.")[1]
    if "part_adaptor" in name:
        return tail.startswith(("0.0.", "0.2.", "1.0."))
    return tail.startswith(("0.", "1."))


def fill_state_dict(schema: dict, seed: int = 0, mode: str = "stress", device="cpu", include_track: bool = False) -> dict:
    """schema: {name: {"shape": [...], "dtype": "torch.float32"}} -> {name: tensor} for every
    floating-point entry (integer buffers are left to the module that owns them).  `track_head.*` is filled only on
    request: it runs only when query_points is given (vggt.py:220) and costs 0.1 B hash draws otherwise."""
    out = {}
    for name, meta in schema.items():
        if not meta["dtype"].startswith("torch.float"):
            continue
        if name.startswith("track_head.") and not include_track:
            continue
        out[name] = make_tensor(name, meta["shape"], seed, mode, device)
    return out


def make_images(S: int, H: int, W: int, seed: int = 1, device="cpu", mode: str = "noise") -> torch.Tensor:
    """Synthetic views in [0,1), bit-identical on CPU and GPU.

    mode "noise" (default; what every hash-noise fixture under tests/golden was generated with): iid hash noise per
    pixel and channel.  White noise makes the patch tokens statistically homogeneous -- the easiest input for the
    mean-input compensation and the static softmax bound, which is why tests/test_real_images_gpu.py runs photographs.
    mode "blocks": piecewise-constant 16 x 16 pixel colour blocks (one hash draw per block and channel; a few dark / bright
    regions per view) + 25 % hash noise: low-frequency structure with edges, for tests that need inhomogeneous tokens at
    arbitrary sizes.  k / 256-grid values and one exact fp32 multiply-add each, so CPU and GPU agree bit for bit."""
    n = S * 3 * H * W
    noise = hash_uniform(n, _name_seed("images", seed), device).view(S, 3, H, W)
    if mode == "noise":
        return noise
    if mode != "blocks":
        raise ValueError(f"unknown image mode {mode!r}")
    bh, bw = (H + 15) // 16, (W + 15) // 16
    coarse = hash_uniform(S * 3 * bh * bw, _name_seed("image_blocks", seed), device).view(S, 3, bh, bw)
    coarse = torch.floor(coarse * 256.0) * (1.0 / 256.0)                 # 8-bit levels: the blend below is exact in fp32
    blocks = coarse.repeat_interleave(16, 2).repeat_interleave(16, 3)[:, :, :H, :W]
    q = torch.floor(noise * 256.0) * (1.0 / 256.0)
    return (blocks * 0.75).add_(q * 0.25)
