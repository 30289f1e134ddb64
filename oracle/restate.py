"""CPU restatement of the reference IGGT forward path -- TEST INFRASTRUCTURE, not product code.

Plain PyTorch fp32 on the CPU, functional over a state dict `sd` keyed exactly like the reference
checkpoint.  Each function cites the reference file:line it restates.  Pinned by
tests/test_oracle_golden.py against tests/golden/*.pt (outputs of the reference's own modules,
oracle/make_golden.py).  Used as (a) the checker for sizes without a fixture, (b) `smoke()`,
(c) the `cpu_baseline` leg of bench.py (kind "port").  Never imported by iggt_official_amd/.
"""
import math

import torch
import torch.nn.functional as F

LAYERS = (4, 11, 17, 23)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _convT(sd, p, x, stride, padding=0):
    return F.conv_transpose2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


# ------------------------------------------------------------------------------------------------
# trunk
# ------------------------------------------------------------------------------------------------
def rope2d(t, pos, base=100.0):
    """iggt/layers/rope.py:119-188.  t [B,h,N,64], pos [B,N,2] (y,x) int64."""
    half = t.shape[-1] // 2
    inv = 1.0 / (base ** (torch.arange(0, half, 2).float() / half))             # rope.py:103-104
    n = int(pos.max()) + 1
    ang = torch.einsum("i,j->ij", torch.arange(n, dtype=inv.dtype), inv)         # rope.py:107-108
    ang = torch.cat((ang, ang), -1)                                              # rope.py:112
    cos_t, sin_t = ang.cos(), ang.sin()
    out = []
    for d, x in enumerate(t.chunk(2, -1)):                                       # rope.py:181-185
        c = F.embedding(pos[..., d], cos_t)[:, None]
        s = F.embedding(pos[..., d], sin_t)[:, None]
        rot = torch.cat((-x[..., half // 2:], x[..., :half // 2]), -1)           # rope.py:119-131
        out.append(x * c + rot * s)
    return torch.cat(out, -1)


def attention(sd, p, x, heads, pos=None):
    """iggt/layers/attention.py:50-77."""
    B, N, C = x.shape
    qkv = _lin(sd, p + ".qkv", x).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    if p + ".q_norm.weight" in sd:
        q, k = _ln(sd, p + ".q_norm", q), _ln(sd, p + ".k_norm", k)
    if pos is not None:
        q, k = rope2d(q, pos), rope2d(k, pos)
    o = F.scaled_dot_product_attention(q, k, v)
    return _lin(sd, p + ".proj", o.transpose(1, 2).reshape(B, N, C))


def block(sd, p, x, heads, pos=None, eps=1e-5):
    """iggt/layers/block.py:81-107 (inference branch 105-106) with LayerScale (layer_scale.py:26)."""
    x = x + sd[p + ".ls1.gamma"] * attention(sd, p + ".attn", _ln(sd, p + ".norm1", x, eps), heads, pos)
    h = _lin(sd, p + ".mlp.fc2", F.gelu(_lin(sd, p + ".mlp.fc1", _ln(sd, p + ".norm2", x, eps))))
    return x + sd[p + ".ls2.gamma"] * h


def dino_patch_tokens(sd, images):
    """iggt/layers/vision_transformer.py:183-236,262-281 (ViT-L/14, 4 registers, LN eps 1e-6).
    images: normalised [S,3,H,W] -> x_norm_patchtokens [S, g2, 1024]."""
    p = "aggregator.patch_embed"
    S, _, H, W = images.shape
    x = F.conv2d(images, sd[p + ".patch_embed.proj.weight"], sd[p + ".patch_embed.proj.bias"], stride=14)
    gh, gw = x.shape[-2:]
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd[p + ".cls_token"].expand(S, -1, -1), x), 1)
    pe = sd[p + ".pos_embed"].float()
    N = pe.shape[1] - 1
    if not (gh * gw == N and H == W):                                            # vision_transformer.py:187-215
        M = int(math.sqrt(N))
        patch = F.interpolate(pe[:, 1:].reshape(1, M, M, -1).permute(0, 3, 1, 2), mode="bicubic", antialias=True,
                              size=(gh, gw))
        pe = torch.cat((pe[:, :1], patch.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)), 1)
    x = x + pe
    x = torch.cat((x[:, :1], sd[p + ".register_tokens"].expand(S, -1, -1), x[:, 1:]), 1)
    for i in range(24):
        x = block(sd, f"{p}.blocks.{i}", x, 16, None, 1e-6)
    return _ln(sd, p + ".norm", x, 1e-6)[:, 5:]


def aggregator(sd, images, keep=LAYERS, n_blocks=24):
    """iggt/models/aggregator.py:186-361.  images [S,3,H,W] in [0,1] -> {layer: [1,S,P,2048]}."""
    S, _, H, W = images.shape
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    patch = dino_patch_tokens(sd, (images - mean) / std)                          # aggregator.py:206-213
    gh, gw = H // 14, W // 14

    def special(t):                                                              # aggregator.py:338-361
        t = sd[t]
        return torch.cat([t[:, 0:1], t[:, 1:].expand(1, S - 1, *t.shape[2:])], 1)[0] if S > 1 else t[0, 0:1]

    tokens = torch.cat([special("aggregator.camera_token"), special("aggregator.register_token"), patch], 1)
    P = tokens.shape[1]
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    pos = torch.cat([torch.zeros(5, 2, dtype=torch.long), torch.stack([ys.flatten(), xs.flatten()], -1) + 1], 0)
    pos = pos[None].expand(S, -1, -1)                                            # aggregator.py:236-245
    out = {}
    for i in range(n_blocks):                                                    # aggregator.py:254-270
        tokens = block(sd, f"aggregator.frame_blocks.{i}", tokens.view(S, P, -1), 16, pos)
        fr = tokens
        tokens = block(sd, f"aggregator.global_blocks.{i}", tokens.reshape(1, S * P, -1), 16,
                       pos.reshape(1, S * P, 2))
        if i in keep:
            out[i] = torch.cat([fr.view(1, S, P, -1), tokens.view(1, S, P, -1)], -1)
    return out


# ------------------------------------------------------------------------------------------------
# heads
# ------------------------------------------------------------------------------------------------
def pos_embed(C, h, w, W, H, ratio=0.1):
    """iggt/heads/utils.py:11-108 + dpt_head.py:274-284 -> [1,C,h,w]."""
    a = W / H
    diag = (a * a + 1.0) ** 0.5
    sx, sy = a / diag, 1.0 / diag
    xs = torch.linspace(-sx * (w - 1) / w, sx * (w - 1) / w, w)
    ys = torch.linspace(-sy * (h - 1) / h, sy * (h - 1) / h, h)
    uu, vv = torch.meshgrid(xs, ys, indexing="xy")
    om = 1.0 / 100.0 ** (torch.arange(C // 4, dtype=torch.double) / (C // 4))
    parts = []
    for g in (uu, vv):
        o = torch.einsum("m,d->md", g.reshape(-1), om)
        parts += [o.sin(), o.cos()]
    return (torch.cat(parts, 1).float().view(h, w, C) * ratio).permute(2, 0, 1)[None]


def _token_maps(sd, p, toks, gh, gw, W, H, use_pos):
    """dpt_head.py:225-240 / adaptor.py:206-219: LN(2048) on patch tokens -> NCHW -> 1x1 conv (+pos)."""
    maps = []
    for i, li in enumerate(LAYERS):
        x = toks[li][0, :, 5:]
        x = _ln(sd, p + ".norm", x).permute(0, 2, 1).reshape(x.shape[0], -1, gh, gw)
        x = _conv(sd, f"{p}.projects.{i}", x)
        if use_pos:
            x = x + pos_embed(x.shape[1], gh, gw, W, H)
        maps.append(x)
    return maps


def _rcu(sd, p, x):
    """ResidualConvUnit with in-place ReLU: conv2(relu(conv1(relu x))) + relu x (dpt_head.py:369-411)."""
    r = F.relu(x)
    return _conv(sd, p + ".conv2", F.relu(_conv(sd, p + ".conv1", r, padding=1)), padding=1) + r


def _fusion(sd, p, x0, x1=None, size=None):
    """FeatureFusionBlock.forward, dpt_head.py:455-481."""
    y = x0 if x1 is None else x0 + _rcu(sd, p + ".resConfUnit1", x1)
    y = _rcu(sd, p + ".resConfUnit2", y)
    if size is None:
        size = (y.shape[-2] * 2, y.shape[-1] * 2)
    y = F.interpolate(y, size=tuple(size), mode="bilinear", align_corners=True)
    return _conv(sd, p + ".out_conv", y)


def dpt_head(sd, p, toks, H, W, out_dim_act):
    """DPTHead._forward_impl + scratch_forward (dpt_head.py:192-316) + activate_head (head_act.py:61-125).
    out_dim_act: "inv_log" (point) | "exp" (depth).  Returns preds [1,S,H,W,c], conf [1,S,H,W], (out2,out3,out4)."""
    gh, gw = H // 14, W // 14
    m = _token_maps(sd, p, toks, gh, gw, W, H, True)
    m[0] = _convT(sd, p + ".resize_layers.0", m[0], 4)
    m[1] = _convT(sd, p + ".resize_layers.1", m[1], 2)
    m[3] = _conv(sd, p + ".resize_layers.3", m[3], 2, 1)
    r = [F.conv2d(m[i], sd[f"{p}.scratch.layer{i + 1}_rn.weight"], None, padding=1) for i in range(4)]
    out4 = _fusion(sd, p + ".scratch.refinenet4", r[3], None, r[2].shape[2:])
    out3 = _fusion(sd, p + ".scratch.refinenet3", out4, r[2], r[1].shape[2:])
    out2 = _fusion(sd, p + ".scratch.refinenet2", out3, r[1], r[0].shape[2:])
    out1 = _fusion(sd, p + ".scratch.refinenet1", out2, r[0])
    x = _conv(sd, p + ".scratch.output_conv1", out1, padding=1)
    x = F.interpolate(x, size=(gh * 14, gw * 14), mode="bilinear", align_corners=True)
    x = x + pos_embed(x.shape[1], x.shape[2], x.shape[3], W, H)
    x = _conv(sd, p + ".scratch.output_conv2.2", F.relu(_conv(sd, p + ".scratch.output_conv2.0", x, padding=1)))
    f = x.permute(0, 2, 3, 1)
    xyz, c = f[..., :-1], f[..., -1]
    pts = torch.sign(xyz) * torch.expm1(xyz.abs()) if out_dim_act == "inv_log" else torch.exp(xyz)
    return pts[None], (1 + c.exp())[None], (out2, out3, out4)


def _projects(sd, p, x):
    """adaptor.py:9-35."""
    x = F.relu(_bn(sd, p + ".input_proj.1", F.conv2d(x, sd[p + ".input_proj.0.weight"])))
    y = F.relu(_bn(sd, p + ".residual_conv.1", F.conv2d(x, sd[p + ".residual_conv.0.weight"], padding=1)))
    y = _bn(sd, p + ".residual_conv.4", F.conv2d(y, sd[p + ".residual_conv.3.weight"], padding=1))
    return _conv(sd, p + ".output_proj", y + x)


def sam_projector(sd, toks, H, W):
    """SamProjector.forward, adaptor.py:187-226 (resize stacks 152-175) -> [res1..res4]."""
    p = "part_adaptor"
    m = _token_maps(sd, p, toks, H // 14, W // 14, W, H, False)
    r = p + ".resize_layers"
    a = _projects(sd, r + ".0.1", _convT(sd, r + ".0.0", m[0], 2, 1))
    a = _projects(sd, r + ".0.3", _convT(sd, r + ".0.2", a, 2, 1))
    b = _projects(sd, r + ".1.1", _convT(sd, r + ".1.0", m[1], 2))
    c = _projects(sd, r + ".2.1", m[2])
    d = _projects(sd, r + ".3.1", _conv(sd, r + ".3.0", m[3], 2, 1))
    return [a, b, c, d]


def _mha(q, k, v, heads, scale, bias=None):
    B, Nq, C = q.shape
    d = C // heads
    q = q.view(B, Nq, heads, d).transpose(1, 2) * scale
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    a = q @ k.transpose(-2, -1)
    if bias is not None:
        a = a + bias
    return (a.softmax(-1) @ v).transpose(1, 2).reshape(B, Nq, C)


def cross_attention(sd, p, q, kv, heads=8):
    """heads/block.py:212-242 (explicit softmax branch)."""
    C = q.shape[-1]
    o = _mha(_lin(sd, p + ".projq", q), _lin(sd, p + ".projk", kv), _lin(sd, p + ".projv", kv), heads,
             (C // heads) ** -0.5)
    return _lin(sd, p + ".proj", o)


def _win_part(x, ws):
    """window_sa.py:71-75."""
    b, h, w, c = x.shape
    return x.view(b, h // ws, ws, w // ws, ws, c).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, c)


def _win_rev(wins, ws, h, w):
    """window_sa.py:77-81."""
    b = int(wins.shape[0] / (h * w / ws / ws))
    return wins.view(b, h // ws, w // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(b, h, w, -1)


def _swin_tail(sd, p, body, x):
    """window_sa.py:429-435 / 538-545: conv_after_body + skip, conv 3x3 -> LeakyReLU(0.01) -> conv 3x3."""
    x = _conv(sd, p + ".conv_after_body", body, padding=1) + x
    x = F.leaky_relu(_conv(sd, p + ".conv_before_upsample.0", x, padding=1), 0.01)
    return _conv(sd, p + ".conv_last", x, padding=1)


def swin_sa(sd, p, x):
    """SwinSA.forward + HAB.forward (window_sa.py:417-435, 200-227); x NCHW -> NCHW."""
    b, c, h, w = x.shape
    t = _ln(sd, p + ".patch_embed.norm", x.flatten(2).transpose(1, 2))
    a = p + ".atten_block"
    y = _ln(sd, a + ".norm1", t).view(b, h, w, c)
    yc = y.permute(0, 3, 1, 2)
    cab = _conv(sd, a + ".conv_block.cab.2", F.gelu(_conv(sd, a + ".conv_block.cab.0", yc, padding=1)), padding=1)
    gate = torch.sigmoid(_conv(sd, a + ".conv_block.cab.3.attention.3",
                               F.relu(_conv(sd, a + ".conv_block.cab.3.attention.1", cab.mean((2, 3), keepdim=True)))))
    conv_x = (cab * gate).permute(0, 2, 3, 1).reshape(b, h * w, c)
    win = _win_part(y, 8).view(-1, 64, c)
    qkv = _lin(sd, a + ".attn.qkv", win).view(-1, 64, 3, c)
    o = _mha(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 4, (c // 4) ** -0.5)
    att = _win_rev(_lin(sd, a + ".attn.proj", o).view(-1, 8, 8, c), 8, h, w).view(b, h * w, c)
    t = t + att + conv_x * 0.01
    t = t + _lin(sd, a + ".mlp.fc2", F.gelu(_lin(sd, a + ".mlp.fc1", _ln(sd, a + ".norm2", t))))
    body = _ln(sd, p + ".norm", t).transpose(1, 2).reshape(b, c, h, w)
    return _swin_tail(sd, p, body, x)


def swin_ca(sd, p, x, kv, rpi):
    """SwinCA.forward + OCAB.forward (window_sa.py:525-545, 270-319) incl. the NCHW query-window quirk."""
    b, c, h, w = x.shape
    tn = lambda z: _ln(sd, p + ".patch_embed.norm", z.flatten(2).transpose(1, 2))  # noqa: E731
    tx, tk = tn(x), tn(kv)
    a = p + ".atten_block"
    q = _lin(sd, a + ".q", _ln(sd, a + ".norm1", tx).view(b, h, w, c)).permute(0, 3, 1, 2)
    k = _lin(sd, a + ".k", _ln(sd, a + ".norm1", tk).view(b, h, w, c)).permute(0, 3, 1, 2)
    v = _lin(sd, a + ".v", _ln(sd, a + ".norm1", tk).view(b, h, w, c)).permute(0, 3, 1, 2)
    qw = _win_part(q, 8).view(-1, 64, c)                                         # window_sa.py:286-287 (quirk)
    kvw = F.unfold(torch.cat((k, v), 1), kernel_size=12, stride=8, padding=2)    # window_sa.py:289
    nw = kvw.shape[-1]
    kvw = kvw.view(b, 2, c, 144, nw).permute(1, 0, 4, 3, 2).reshape(2, b * nw, 144, c)
    bias = sd[a + ".relative_position_bias_table"][rpi.view(-1)].view(64, 144, 4).permute(2, 0, 1)[None]
    o = _mha(qw, kvw[0], kvw[1], 4, (c // 4) ** -0.5, bias)
    t = _lin(sd, a + ".proj", _win_rev(o.view(-1, 8, 8, c), 8, h, w).view(b, h * w, c)) + tx
    t = t + _lin(sd, a + ".mlp.fc2", F.gelu(_lin(sd, a + ".mlp.fc1", _ln(sd, a + ".norm2", t))))
    body = _ln(sd, p + ".norm", t).transpose(1, 2).reshape(b, c, h, w)
    return _swin_tail(sd, p, body, x)


def part_head(sd, pyr, point_feat, H, W, rpi_oca):
    """PartHead.scratch_forward + _forward_impl (part_head.py:148-243); cross_attention_1 is dead code
    in the reference (178-185) and skipped.  -> [1,S,8,H,W]."""
    p = "part_head"
    r = [F.conv2d(pyr[i], sd[f"{p}.scratch.layer{i + 1}_rn.weight"], None, padding=1) for i in range(4)]
    out = _fusion(sd, p + ".scratch.refinenet4", r[3], None, r[2].shape[2:])
    q = out.flatten(2).permute(0, 2, 1)
    kv = point_feat[2].flatten(2).permute(0, 2, 1)
    out4 = cross_attention(sd, p + ".cross_attention_2", q, kv).permute(0, 2, 1).reshape(out.shape)
    out = _fusion(sd, p + ".scratch.refinenet3", out4, r[2], r[1].shape[2:])
    out = _fusion(sd, p + ".scratch.refinenet2", out, r[1], r[0].shape[2:])
    out2 = swin_ca(sd, p + ".window_cross_attention", out, point_feat[0], rpi_oca)
    out = _fusion(sd, p + ".scratch.refinenet1", out2, r[0])
    out = _conv(sd, p + ".scratch.output_conv1", out, padding=1)
    out = swin_sa(sd, p + ".window_self_atten", out)
    out = F.interpolate(out, size=(H // 14 * 14, W // 14 * 14), mode="bilinear", align_corners=True)
    out = _conv(sd, p + ".scratch.output_conv2.2", F.relu(_conv(sd, p + ".scratch.output_conv2.0", out, padding=1)))
    return out[None]


def camera_head(sd, cam_tokens, iters=4):
    """CameraHead.forward / trunk_fn (camera_head.py:83-154).  cam_tokens [1,S,2048] -> list of [1,S,9]."""
    p = "camera_head"
    x0 = _ln(sd, p + ".token_norm", cam_tokens)
    pred, outs = None, []
    for _ in range(iters):
        inp = sd[p + ".empty_pose_tokens"].expand(*x0.shape[:2], -1) if pred is None else pred
        mod = _lin(sd, p + ".poseLN_modulation.1", F.silu(_lin(sd, p + ".embed_pose", inp)))
        shift, scale, gate = mod.chunk(3, -1)
        x = gate * (F.layer_norm(x0, (x0.shape[-1],), None, None, 1e-6) * (1 + scale) + shift) + x0
        for i in range(4):
            x = block(sd, f"{p}.trunk.{i}", x, 16)
        d = _lin(sd, p + ".pose_branch.fc2", F.gelu(_lin(sd, p + ".pose_branch.fc1", _ln(sd, p + ".trunk_norm", x))))
        pred = d if pred is None else pred + d
        outs.append(torch.cat([pred[..., :7], F.relu(pred[..., 7:])], -1))         # head_act.py:11-32 (fl relu)
    return outs


def rpi_oca(ws=8, ratio=0.5):
    """window_sa.py:500-523."""
    we = ws + int(ratio * ws)
    co = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    ce = torch.stack(torch.meshgrid([torch.arange(we), torch.arange(we)], indexing="ij")).flatten(1)
    rel = (ce[:, None, :] - co[:, :, None]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - we + 1
    rel[:, :, 1] += ws - we + 1
    rel[:, :, 0] *= ws + we - 1
    return rel.sum(-1)


@torch.no_grad()
def iggt_forward(sd, images, with_part=None):
    """IGGT.forward orchestration, vggt.py:185-218.  images [S,3,H,W] in [0,1], CPU fp32."""
    S, _, H, W = images.shape
    toks = aggregator(sd, images)
    out = {"tokens": toks, "pose_enc": camera_head(sd, toks[23][:, :, 0])}
    out["depth"], out["depth_conf"], _ = dpt_head(sd, "depth_head", toks, H, W, "exp")
    out["world_points"], out["world_points_conf"], pf = dpt_head(sd, "point_head", toks, H, W, "inv_log")
    out["point_feat"] = pf
    if with_part is None:
        with_part = (H % 28 == 0 and W % 28 == 0)
    if with_part:
        pyr = sam_projector(sd, toks, H, W)
        out["adaptor"] = pyr
        out["part_feat"] = part_head(sd, pyr, pf, H, W, rpi_oca())
    return out
