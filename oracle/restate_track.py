"""CPU restatement of the reference track head (the `query_points` path) -- TEST INFRASTRUCTURE, not product code.

Plain PyTorch fp32, functional over a state dict keyed like the reference checkpoint (`track_head.*`).  Pinned by
tests/test_oracle_golden.py against tests/golden/track_*.pt (outputs of the reference's own TrackHead,
oracle/make_golden_track.py).  Used as the checker of the HIP track head at sizes without a fixture.  Never imported by
iggt_official_amd/.
"""
import math

import torch
import torch.nn.functional as F

from .restate import _conv, _convT, _fusion, _lin, _ln, _token_maps

P = "track_head.tracker"


def track_features(sd, toks, H, W):
    """TrackHead.feature_extractor = DPTHead(features=128, for_tracker=True, down_ratio=2, pos_embed=False)
    (track_head.py:50-59; dpt_head.py:192-262): [1, S, 128, H/2, W/2]."""
    p = "track_head.feature_extractor"
    gh, gw = H // 14, W // 14
    m = _token_maps(sd, p, toks, gh, gw, W, H, False)
    m[0] = _convT(sd, p + ".resize_layers.0", m[0], 4)
    m[1] = _convT(sd, p + ".resize_layers.1", m[1], 2)
    m[3] = _conv(sd, p + ".resize_layers.3", m[3], 2, 1)
    r = [F.conv2d(m[i], sd[f"{p}.scratch.layer{i + 1}_rn.weight"], None, padding=1) for i in range(4)]
    out4 = _fusion(sd, p + ".scratch.refinenet4", r[3], None, r[2].shape[2:])
    out3 = _fusion(sd, p + ".scratch.refinenet3", out4, r[2], r[1].shape[2:])
    out2 = _fusion(sd, p + ".scratch.refinenet2", out3, r[1], r[0].shape[2:])
    out1 = _fusion(sd, p + ".scratch.refinenet1", out2, r[0])
    x = _conv(sd, p + ".scratch.output_conv1", out1, padding=1)
    x = F.interpolate(x, size=(int(gh * 14 / 2), int(gw * 14 / 2)), mode="bilinear", align_corners=True)
    return x[None]


def _sample(img, xy, padding):
    """track_modules/utils.py:124-189 `bilinear_sampler`, align_corners=True: img [B,C,H,W], xy [B,h,w,2] in pixels."""
    H, W = img.shape[-2:]
    g = xy * torch.tensor([2.0 / max(W - 1, 1), 2.0 / max(H - 1, 1)]) - 1.0
    return F.grid_sample(img, g, align_corners=True, padding_mode=padding)


def sample_points(img, xy):
    """utils.py:192-226 `sample_features4d`: img [B,C,H,W], xy [B,R,2] -> [B,R,C] (border padding)."""
    return _sample(img, xy[:, :, None], "border")[..., 0].permute(0, 2, 1)


def sincos_grid(dim, hh, ww):
    """utils.py:17-87 `get_2d_sincos_pos_embed`: [1, dim, hh, ww]; first half of the channels encodes x, second half y,
    each as [sin | cos] of pos * 10000^(-j / (dim/4)) evaluated in fp64."""
    q = dim // 4
    om = 1.0 / 10000 ** (torch.arange(q, dtype=torch.double) / q)
    gx = torch.arange(ww, dtype=torch.float).double()[:, None] * om          # [ww, q]
    gy = torch.arange(hh, dtype=torch.float).double()[:, None] * om
    ex = torch.cat([gx.sin(), gx.cos()], 1).float()                          # [ww, dim/2]
    ey = torch.cat([gy.sin(), gy.cos()], 1).float()
    emb = torch.cat([ex[None].expand(hh, ww, -1), ey[:, None].expand(hh, ww, -1)], -1)
    return emb.permute(2, 0, 1)[None]


def flow_embedding(flows, C):
    """utils.py:90-121 `get_2d_embedding(cat_coords=False)`: flows [M,S,2] -> [M,S,2C], sin / cos interleaved."""
    div = torch.arange(0, C, 2, dtype=torch.float32) * (1000.0 / C)
    out = []
    for a in (flows[..., 0:1], flows[..., 1:2]):
        ang = a * div
        out.append(torch.stack([ang.sin(), ang.cos()], -1).flatten(-2))
    return torch.cat(out, -1)


def corr_pyramid(fmaps, levels):
    """blocks.py:170-180: level l = l times 2x2 average pooling (floor sizes)."""
    pyr = [fmaps]
    for _ in range(levels - 1):
        pyr.append(F.avg_pool2d(pyr[-1], 2, 2))
    return pyr


def corr_sample(pyr, targets, coords, radius):
    """blocks.py:189-241 `CorrBlock.corr_sample`: pyr[l] [S,C,H_l,W_l], targets [S,N,C], coords [S,N,2] (level-0
    pixels) -> [S,N,levels*(2r+1)^2]: the full correlation map target . fmap / sqrt(C) of every level, sampled
    bilinearly (zero padding) on the (2r+1)^2 integer offsets around coords / 2^l; offsets vary first in y."""
    S, N, C = targets.shape
    d = torch.linspace(-radius, radius, 2 * radius + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), -1)             # [.., 0] varies along dim 0
    out = []
    for l, fm in enumerate(pyr):
        H, W = fm.shape[-2:]
        corr = torch.matmul(targets, fm.reshape(S, C, H * W)) / math.sqrt(C)   # [S,N,HW]
        grid = coords.reshape(S * N, 1, 1, 2) / 2 ** l + delta[None]
        smp = _sample(corr.reshape(S * N, 1, H, W), grid, "zeros")
        out.append(smp.reshape(S, N, -1))
    return torch.cat(out, -1)


def _mha(sd, p, q_in, kv_in, heads=8):
    """nn.MultiheadAttention(batch_first=True) forward, eval mode: packed in_proj, softmax(q k^T / sqrt(d)) v, out_proj."""
    E = q_in.shape[-1]
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(q_in, w[:E], b[:E])
    k = F.linear(kv_in, w[E:2 * E], b[E:2 * E])
    v = F.linear(kv_in, w[2 * E:], b[2 * E:])
    B, Nq, _ = q.shape
    d = E // heads

    def split(t):
        return t.reshape(B, -1, heads, d).transpose(1, 2)

    a = torch.softmax(split(q) @ split(k).transpose(-1, -2) / math.sqrt(d), -1) @ split(v)
    return _lin(sd, p + ".out_proj", a.transpose(1, 2).reshape(B, Nq, E))


def _mlp(sd, p, x):
    return _lin(sd, p + ".fc2", F.gelu(_lin(sd, p + ".fc1", x)))


def attn_block(sd, p, x):
    """modules.py:176-192: NB the residual is taken from the NORMALISED input."""
    x = _ln(sd, p + ".norm1", x)
    x = x + _mha(sd, p + ".attn", x, x)
    return x + _mlp(sd, p + ".mlp", _ln(sd, p + ".norm2", x))


def cross_block(sd, p, x, ctx):
    """modules.py:206-218."""
    x = _ln(sd, p + ".norm1", x)
    ctx = _ln(sd, p + ".norm_context", ctx)
    x = x + _mha(sd, p + ".cross_attn", x, ctx)
    return x + _mlp(sd, p + ".mlp", _ln(sd, p + ".norm2", x))


def update_former(sd, x, depth=6, n_virtual=64):
    """blocks.py:100-143 `EfficientUpdateFormer.forward`: x [N,S,388] -> [N,S,130] (B = 1)."""
    p = P + ".updateformer"
    tok = _lin(sd, p + ".input_transform", _ln(sd, p + ".input_norm", x))
    init = tok
    N, S, C = tok.shape
    tok = torch.cat([tok, sd[p + ".virual_tracks"][0].expand(n_virtual, S, C)], 0)       # [N+V,S,C]
    for i in range(depth):
        tok = attn_block(sd, f"{p}.time_blocks.{i}", tok)                       # attention along S, per track
        sp = tok.permute(1, 0, 2)                                               # [S,N+V,C]: attention along tracks
        pt, vt = sp[:, :N], sp[:, N:]
        vt = cross_block(sd, f"{p}.space_virtual2point_blocks.{i}", vt, pt)
        vt = attn_block(sd, f"{p}.space_virtual_blocks.{i}", vt)
        pt = cross_block(sd, f"{p}.space_point2virtual_blocks.{i}", pt, vt)
        tok = torch.cat([pt, vt], 1).permute(1, 0, 2)
    tok = tok[:N] + init
    return _lin(sd, p + ".flow_head", _ln(sd, p + ".output_norm", tok))


def tracker_init(sd, fmaps, query_points, stride=2, levels=7, latent=128):
    """base_track_predictor.py:87-126: everything before the refinement loop (B = 1).  State (frame-major, [S,N,.])."""
    _, S, C, HH, WW = fmaps.shape
    fm = _ln(sd, P + ".fmap_norm", fmaps[0].permute(0, 2, 3, 1)).permute(0, 3, 1, 2)      # [S,C,HH,WW]
    coords = (query_points[0] / float(stride))[None].repeat(S, 1, 1)                      # [S,N,2]
    qfeat = sample_points(fm[:1], coords[:1])[0]                                          # [N,C]
    tdim = 3 * latent + 4
    ref = sd[P + ".query_ref_token"][0]                                                   # [2,tdim]
    return dict(S=S, N=query_points.shape[1], coords=coords, coords0=coords.clone(),
                feats=qfeat[None].repeat(S, 1, 1), pyr=corr_pyramid(fm, levels),
                pos=sample_points(sincos_grid(tdim, HH, WW), coords[:1])[0],              # [N,tdim]
                ref=torch.cat([ref[:1], ref[1:2].expand(S - 1, -1)], 0))                  # [S,tdim]


def tracker_step(sd, st, stride=2, radius=4, latent=128, max_scale=518, taps=None):
    """One refinement iteration, base_track_predictor.py:129-195; updates st["coords"], st["feats"], returns the
    prediction [1,S,N,2] in image pixels."""
    S, N, coords, feats = st["S"], st["N"], st["coords"], st["feats"]
    fc = corr_sample(st["pyr"], feats, coords, radius).permute(1, 0, 2)                   # [N,S,567]
    if taps is not None:
        taps["fcorrs"] = fc.clone()
    fc = _mlp(sd, P + ".corr_mlp", fc)
    flows = (coords - coords[:1]).permute(1, 0, 2)                                        # [N,S,2]
    femb = torch.cat([flow_embedding(flows, latent // 2), flows / max_scale, flows / max_scale], -1)
    tf = feats.permute(1, 0, 2)                                                           # [N,S,C]
    x = torch.cat([femb, fc, tf], -1) + st["pos"][:, None] + st["ref"][None]
    delta = update_former(sd, x)
    if taps is not None:
        taps["delta"] = delta[None].clone()
    dfeat = delta[..., 2:].reshape(N * S, latent)
    upd = F.gelu(_lin(sd, P + ".ffeat_updater.0",
                      F.group_norm(dfeat, 1, sd[P + ".ffeat_norm.weight"], sd[P + ".ffeat_norm.bias"])))
    st["feats"] = (upd + tf.reshape(N * S, latent)).reshape(N, S, latent).permute(1, 0, 2)
    coords = coords + delta[..., :2].permute(1, 0, 2)
    coords[0] = st["coords0"][0]
    st["coords"] = coords
    return (coords * stride)[None].clone()


def tracker_heads(sd, st, latent=128):
    """base_track_predictor.py:197-207 -> vis, conf [1,S,N]."""
    f2 = st["feats"].reshape(st["S"] * st["N"], latent)
    vis = torch.sigmoid(_lin(sd, P + ".vis_predictor.0", f2)).reshape(1, st["S"], st["N"])
    conf = torch.sigmoid(_lin(sd, P + ".conf_predictor.0", f2)).reshape(1, st["S"], st["N"])
    return vis, conf


def tracker(sd, fmaps, query_points, iters=4, taps=None):
    """base_track_predictor.py:87-209 `BaseTrackerPredictor.forward` for B = 1: fmaps [1,S,C,HH,WW], query_points
    [1,N,2] (image pixels) -> (list of iters x [1,S,N,2], vis [1,S,N], conf [1,S,N])."""
    st = tracker_init(sd, fmaps, query_points)
    preds = []
    for it in range(iters):
        t = {} if (taps is not None and it == 0) else None
        preds.append(tracker_step(sd, st, taps=t))
        if t:
            taps["fcorrs_it0"], taps["delta_it0"] = t["fcorrs"], t["delta"]
    vis, conf = tracker_heads(sd, st)
    return preds, vis, conf


def track_head(sd, toks, H, W, query_points, iters=4, taps=None):
    """TrackHead.forward (track_head.py:75-109)."""
    fmaps = track_features(sd, toks, H, W)
    if taps is not None:
        taps["fmaps"] = fmaps
    return tracker(sd, fmaps, query_points, iters=iters, taps=taps)
