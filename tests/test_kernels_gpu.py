"""Per-kernel parity tests (GPU): every HIP kernel, called through the C ABI, against a plain
PyTorch fp64 reference of the same op evaluated on the SAME bf16-rounded operands, so the only
admissible differences are accumulation order and the kernel's documented internal roundings
(P and O in bf16 for attention, bf16 outputs).  Tolerances are stated per test."""
import pytest
import torch

from conftest import report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def C():
    from iggt_official_amd import _C

    _C.load()
    return _C


def _rand(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


def _relerr(a, b):
    a, b = a.double(), b.double()
    return (float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)),
            float((a - b).norm() / b.norm().clamp_min(1e-30)))


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 1024), (1374, 3072, 1024), (77, 1024, 4096),
                                   (1000, 96, 640), (33, 9, 128)])
def test_gemm_plain_f32_out(C, M, N, K):
    a = _rand((M, K), 1, dtype=torch.bfloat16)
    w = _rand((N, K), 2, K ** -0.5, dtype=torch.bfloat16)
    bias = _rand((N,), 3)
    out = torch.full((M, N), float("nan"), device="cuda")
    C.gemm_bf16(a, w, out, bias=bias)
    ref = a.double() @ w.double().t() + bias.double()
    mx, l2 = _relerr(out, ref)
    report(f"gemm_f32_{M}x{N}x{K}", dict(max=mx, l2=l2))
    assert mx < 2e-5, (mx, l2)  # fp32 accumulation of exact bf16 products


def test_gemm_asymmetric_layout(C):
    """Permutation-like A with an asymmetric W catches a swapped row/col in the C write (guide rule 16)."""
    M = N = 128
    K = 128
    a = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    a[torch.arange(M), torch.arange(M) % K] = 1
    w = (torch.arange(N * K, device="cuda").view(N, K) % 251).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda")
    C.gemm_bf16(a, w, out)
    ref = a.float() @ w.float().t()
    assert torch.equal(out, ref)


def test_gemm_bf16_out_gelu(C):
    M, N, K = 515, 4096, 1024
    a = _rand((M, K), 4, dtype=torch.bfloat16)
    w = _rand((N, K), 5, K ** -0.5, dtype=torch.bfloat16)
    bias = _rand((N,), 6, 0.1)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    C.gemm_bf16(a, w, out, bias=bias, act=1)
    ref = torch.nn.functional.gelu(a.double() @ w.double().t() + bias.double())
    mx, l2 = _relerr(out, ref)
    report("gemm_gelu_bf16", dict(max=mx, l2=l2))
    assert mx < 6e-3 and l2 < 3e-3  # one bf16 rounding of the output (2^-9 relative)


def test_gemm_layerscale_residual_accumulate(C):
    M, N, K = 1374 * 2, 1024, 4096
    a = _rand((M, K), 7, dtype=torch.bfloat16)
    w = _rand((N, K), 8, K ** -0.5, dtype=torch.bfloat16)
    bias, gamma = _rand((N,), 9, 0.1), _rand((N,), 10)
    x = _rand((M, N), 11)
    ref = x.double() + gamma.double() * (a.double() @ w.double().t() + bias.double())
    C.gemm_bf16(a, w, x, bias=bias, gamma=gamma, accumulate=True)
    mx, l2 = _relerr(x, ref)
    assert mx < 2e-5, (mx, l2)


def test_gemm_row_remap_add_table(C):
    """patch-embed epilogue: rows scattered behind 5 special rows per view, + pos table."""
    S, g2, N, K = 3, 37, 1024, 640
    a = _rand((S * g2, K), 12, dtype=torch.bfloat16)
    w = _rand((N, K), 13, K ** -0.5, dtype=torch.bfloat16)
    bias, table = _rand((N,), 14, 0.1), _rand((g2, N), 15)
    out = torch.full((S * (g2 + 5), N), -7.0, device="cuda")
    C.gemm_bf16(a, w, out, bias=bias, add_table=table, rows_in=g2, rows_out=g2 + 5, row_off=5)
    ref = (a.double() @ w.double().t() + bias.double()).view(S, g2, N) + table.double()
    o = out.view(S, g2 + 5, N)
    assert torch.all(o[:, :5] == -7.0)
    mx, _ = _relerr(o[:, 5:], ref)
    assert mx < 2e-5


def test_gemm_strided_a(C):
    M, N, K = 300, 256, 1024
    big = _rand((M, 3 * K), 16, dtype=torch.bfloat16)
    a = big[:, K:2 * K]
    w = _rand((N, K), 17, K ** -0.5, dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda")
    C.gemm_bf16(a, w, out)
    mx, _ = _relerr(out, a.double() @ w.double().t())
    assert mx < 2e-5


# ---------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, scale):
    s = (q.double() @ k.double().transpose(-1, -2)) * scale
    return torch.softmax(s, -1) @ v.double()


@pytest.mark.parametrize("B,H,Nq,Nk,tile", [(1, 1, 64, 64, 5128), (2, 16, 1374, 1374, 0), (1, 16, 4122, 4122, 0),
                                            (3, 4, 21, 21, 0), (1, 16, 2748, 5496, 5256),
                                            (1, 16, 4122, 4122, 5256), (1, 16, 4122, 4122, 6256), (2, 3, 1374, 1374, 6128),
                                            (1, 2, 300, 777, 6256), (1, 2, 300, 777, 5128), (1, 1, 40, 64, 6256),
                                            (1, 2, 1000, 65, 6256), (1, 2, 500, 129, 6128)])
def test_flash_attn_packed_qkv_layout(C, B, H, Nq, Nk, tile):
    """q,k,v read straight out of a [T, 3*C] qkv matrix (row stride 3C), o written as [T, C]."""
    Cdim = H * 64
    N = max(Nq, Nk)
    qkv = _rand((B * N, 3 * Cdim), 20 + Nq, 1.0, torch.bfloat16)
    o = torch.full((B * Nq, Cdim), float("nan"), dtype=torch.bfloat16, device="cuda")
    C.flash_attn_d64(qkv, qkv[:, Cdim:], qkv[:, 2 * Cdim:], o, B, H, Nq, Nk,
                     N * 3 * Cdim, 3 * Cdim, N * 3 * Cdim, 3 * Cdim, N * 3 * Cdim, 3 * Cdim, Nq * Cdim, Cdim,
                     0.125, tile)
    x = qkv.view(B, N, 3, H, 64)
    q, k, v = x[:, :Nq, 0].transpose(1, 2), x[:, :Nk, 1].transpose(1, 2), x[:, :Nk, 2].transpose(1, 2)
    ref = _attn_ref(q, k, v, 0.125).transpose(1, 2).reshape(B * Nq, Cdim)
    assert not torch.isnan(o.float()).any()
    mx, l2 = _relerr(o, ref)
    report(f"attn_B{B}_H{H}_{Nq}x{Nk}_t{tile}", dict(max=mx, l2=l2))
    # P and O are rounded to bf16 (2^-9 each): 1.5e-2 of the output range, 4e-3 in l2
    assert mx < 1.5e-2 and l2 < 4e-3, (mx, l2)


def test_flash_attn_peaked_and_rescale(C):
    """A key whose score dwarfs the rest arrives in a late tile: forces the online-softmax rescale
    (guide rule 26) and a near one-hot softmax."""
    H, N = 2, 1000
    Cdim = H * 64
    qkv = _rand((N, 3 * Cdim), 33, 0.5, torch.bfloat16)
    x = qkv.view(N, 3, H, 64)
    x[700, 1] = x[123, 0] * 8.0   # key 700 aligned with query 123
    x[901, 1] = x[5, 0] * 6.0
    o = torch.empty(N, Cdim, dtype=torch.bfloat16, device="cuda")
    C.flash_attn_d64(qkv, qkv[:, Cdim:], qkv[:, 2 * Cdim:], o, 1, H, N, N, 0, 3 * Cdim, 0, 3 * Cdim, 0, 3 * Cdim,
                     0, Cdim, 0.125, 0)
    q, k, v = x[:, 0].transpose(0, 1), x[:, 1].transpose(0, 1), x[:, 2].transpose(0, 1)
    ref = _attn_ref(q, k, v, 0.125).transpose(0, 1).reshape(N, Cdim)
    mx, l2 = _relerr(o, ref)
    assert mx < 1.5e-2 and l2 < 4e-3, (mx, l2)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Cdim", [1024, 2048, 256])
def test_layernorm(C, Cdim):
    rows = 1003
    x = _rand((rows, Cdim), 40, 3.0) + 0.7
    w, b = _rand((Cdim,), 41) * 0.1 + 1, _rand((Cdim,), 42, 0.1)
    out = torch.empty(rows, Cdim, dtype=torch.bfloat16, device="cuda")
    C.layernorm(x, w, b, out, 1e-5)
    ref = torch.nn.functional.layer_norm(x.double(), (Cdim,), w.double(), b.double(), 1e-5)
    mx, l2 = _relerr(out, ref)
    assert mx < 5e-3 and l2 < 3e-3
    outf = torch.empty(rows, Cdim, device="cuda")
    C.layernorm(x, w, b, outf, 1e-5)
    mx, _ = _relerr(outf, ref)
    assert mx < 1e-5


def test_layernorm_concat_and_row_remap(C):
    S, P, psi, Ch = 3, 21, 5, 1024
    f, g = _rand((S * P, Ch), 43), _rand((S * P, Ch), 44, 2.0)
    w, b = _rand((2 * Ch,), 45) * 0.1 + 1, _rand((2 * Ch,), 46, 0.1)
    g2 = P - psi
    out = torch.empty(S * g2, 2 * Ch, dtype=torch.float32, device="cuda")
    C.layernorm(f, w, b, out, 1e-5, x1=g, rows=S * g2, rows_in=g2, rows_stride=P, row_off=psi)
    cat = torch.cat([f, g], -1).view(S, P, 2 * Ch)[:, psi:].reshape(S * g2, 2 * Ch)
    ref = torch.nn.functional.layer_norm(cat.double(), (2 * Ch,), w.double(), b.double(), 1e-5)
    mx, _ = _relerr(out, ref)
    assert mx < 1e-5
    # output remap: write behind 5 rows of a [S, P, C] buffer
    x = _rand((S, P, Ch), 47)
    dst = torch.full((S, P, Ch), -3.0, device="cuda")
    C.layernorm(x, w[:Ch].contiguous(), b[:Ch].contiguous(), dst, 1e-6, rows=S * g2, rows_in=g2, rows_stride=P,
                row_off=psi, orows_stride=P, orow_off=psi)
    ref = torch.nn.functional.layer_norm(x[:, psi:].double(), (Ch,), w[:Ch].double(), b[:Ch].double(), 1e-6)
    assert torch.all(dst[:, :psi] == -3.0)
    mx, _ = _relerr(dst[:, psi:], ref)
    assert mx < 1e-5


def _rope_ref(t, pos, base=100.0):
    """fp64 restatement of reference rope.py:119-188 on [T, H, 64] with pos [T, 2]."""
    half = 32
    inv = 1.0 / base ** (torch.arange(0, half, 2, dtype=torch.float64) / half)
    out = []
    for d in range(2):
        x = t[..., d * half:(d + 1) * half]
        ang = pos[:, d].double()[:, None] * inv[None]
        ang = torch.cat([ang, ang], -1)[:, None, :]
        rot = torch.cat([-x[..., half // 2:], x[..., :half // 2]], -1)
        out.append(x * ang.cos() + rot * ang.sin())
    return torch.cat(out, -1)


def test_qknorm_rope(C):
    from iggt_official_amd.layers.rope import RotaryPositionEmbedding2D

    S, gh, gw, psi = 2, 5, 7, 5
    P = psi + gh * gw
    T = S * P
    qkv = _rand((T, 3072), 50, 1.5, torch.bfloat16)
    orig = qkv.clone()
    qw, qb, kw, kb = (_rand((64,), 51) * 0.1 + 1, _rand((64,), 52, 0.1), _rand((64,), 53) * 0.1 + 1,
                      _rand((64,), 54, 0.1))
    cos, sin = RotaryPositionEmbedding2D(100).tables(64, max(gh, gw), torch.device("cuda"))
    vcopy = torch.empty(T, 1024, dtype=torch.bfloat16, device="cuda")
    C.qknorm_rope(qkv, qkv, qkv[:, 1024:], vcopy, qw, qb, kw, kb, cos, sin, T, P, gw, psi, 1e-5)
    pos = torch.zeros(P, 2, dtype=torch.long)
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    pos[psi:, 0], pos[psi:, 1] = ys.flatten() + 1, xs.flatten() + 1
    pos = pos.repeat(S, 1)
    for idx, (w_, b_) in enumerate([(qw, qb), (kw, kb)]):
        t = orig[:, idx * 1024:(idx + 1) * 1024].double().cpu().view(T, 16, 64)
        t = torch.nn.functional.layer_norm(t, (64,), w_.double().cpu(), b_.double().cpu(), 1e-5)
        ref = _rope_ref(t, pos).reshape(T, 1024)
        got = qkv[:, idx * 1024:(idx + 1) * 1024].double().cpu()
        mx, l2 = _relerr(got, ref)
        assert mx < 6e-3 and l2 < 3e-3, (idx, mx, l2)  # one bf16 rounding of the result
    assert torch.equal(qkv[:, 2048:], orig[:, 2048:]) and torch.equal(vcopy, orig[:, 2048:])


def test_im2row_and_special_tokens(C):
    S, H, W = 2, 28, 42
    img = torch.rand(S, 3, H, W, generator=torch.Generator().manual_seed(60)).cuda()
    gh, gw = H // 14, W // 14
    out = torch.empty(S * gh * gw, 640, dtype=torch.bfloat16, device="cuda")
    C.im2row_patch14(img, out, S, H, W, 640)
    mean = torch.tensor([0.485, 0.456, 0.406], device="cuda").view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device="cuda").view(1, 3, 1, 1)
    ref = torch.nn.functional.unfold((img - mean) / std, 14, stride=14).transpose(1, 2).reshape(S * gh * gw, 588)
    assert torch.all(out[:, 588:] == 0)
    d = (out[:, :588].float() - ref).abs().max()
    assert d < 2e-2  # bf16 rounding of O(1..2.6) values; division order may differ by 1 ulp of fp32
    dst = torch.zeros(3, 9, 1024, device="cuda")
    a, b = _rand((5, 1024), 61), _rand((5, 1024), 62)
    C.write_special_tokens(dst, a, b, 3, 5, 0, True)
    assert torch.equal(dst[0, :5], a) and torch.equal(dst[1, :5], b) and torch.equal(dst[2, :5], b)
    assert torch.all(dst[:, 5:] == 0)
    C.write_special_tokens(dst, a, b, 3, 5, 0, False)
    assert torch.equal(dst[0, :5], b)


@pytest.mark.parametrize("M,N,K,mode", [(5000, 4096, 1024, "gelu"), (4122, 1024, 4096, "acc"), (2738, 1024, 640, "remap"),
                                        (43968, 3072, 1024, "plain"), (1024, 256, 2048, "plain")])
def test_gemm_large_tile_path(C, M, N, K, mode):
    """Shapes routed to the 256x256 LDS-DMA kernel (M >= 1024, N % 256 == 0): tails, epilogues, remap."""
    a = _rand((M, K), 70, dtype=torch.bfloat16)
    w = _rand((N, K), 71, K ** -0.5, dtype=torch.bfloat16)
    bias, gamma = _rand((N,), 72, 0.1), _rand((N,), 73)
    base = (a.double() @ w.double().t() + bias.double()) if M < 20000 else None
    if mode == "gelu":
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        C.gemm_bf16(a, w, out, bias=bias, act=1)
        mx, l2 = _relerr(out, torch.nn.functional.gelu(base))
        assert mx < 6e-3 and l2 < 3e-3
    elif mode == "acc":
        x = _rand((M, N), 74)
        ref = x.double() + gamma.double() * base
        C.gemm_bf16(a, w, x, bias=bias, gamma=gamma, accumulate=True)
        assert _relerr(x, ref)[0] < 2e-5
    elif mode == "remap":
        g2 = 1369
        S = M // g2
        table = _rand((g2, N), 75)
        out = torch.full((S * (g2 + 5), N), -7.0, device="cuda")
        C.gemm_bf16(a, w, out, bias=bias, add_table=table, rows_in=g2, rows_out=g2 + 5, row_off=5)
        o = out.view(S, g2 + 5, N)
        assert torch.all(o[:, :5] == -7.0)
        assert _relerr(o[:, 5:], base.view(S, g2, N) + table.double())[0] < 2e-5
    else:
        out = torch.full((M, N), float("nan"), device="cuda")
        C.gemm_bf16(a, w, out, bias=bias)
        if base is None:   # too big for an fp64 matmul of the whole thing: check a row sample
            rows = torch.arange(0, M, 97, device="cuda")
            base_s = a[rows].double() @ w.double().t() + bias.double()
            assert not torch.isnan(out).any() and _relerr(out[rows], base_s)[0] < 2e-5
        else:
            assert _relerr(out, base)[0] < 2e-5


@pytest.mark.parametrize("Cout,act,conf_act", [(4, "inv_log", "expp1"), (2, "exp", "expp1"), (4, "norm", "sigmoid"),
                                               (8, "linear", "expp0"), (3, "relu", "expp1"), (4, "sigmoid", "expp1")])
def test_head_tail(C, Cout, act, conf_act):
    """1x1 conv 32 -> Cout + activate_head, fp32 (reference dpt_head.py:121-128, head_act.py:61-125)."""
    from iggt_official_amd.heads.head_act import activate_head

    x = _rand((3, 37, 41, 32), 80, 0.7)
    w, b = _rand((Cout, 32), 81, 0.2), _rand((Cout,), 82, 0.3)
    pts, conf = C.head_tail(x, w, b, act, conf_act)
    lin = torch.nn.functional.linear(x.double(), w.double(), b.double())            # NHWC
    rp, rc = activate_head(lin.permute(0, 3, 1, 2), activation=act, conf_activation=conf_act)
    assert pts.shape == rp.shape and conf.shape == rc.shape
    assert _relerr(pts, rp)[0] < 2e-6 and _relerr(conf, rc)[0] < 2e-6


def test_tokenops_layernorm_and_linear(C):
    """Part-head token ops on HIP: LayerNorm (incl. the 32-lane C = 128 kernel) and Linear as a split-bf16 1x1 GEMM with
    fused GELU / residual, against fp64."""
    import torch.nn as nn

    from iggt_official_amd.heads import tokenops as tk

    for Cdim in (128, 256):
        x = _rand((3, 77, Cdim), 90 + Cdim, 2.0) + 0.3
        norm = nn.LayerNorm(Cdim).cuda()
        with torch.no_grad():
            norm.weight.copy_(_rand((Cdim,), 91) * 0.1 + 1)
            norm.bias.copy_(_rand((Cdim,), 92, 0.1))
        y = tk.layer_norm(norm, x)
        ref = torch.nn.functional.layer_norm(x.double(), (Cdim,), norm.weight.double(), norm.bias.double(), norm.eps)
        assert y.shape == x.shape and _relerr(y, ref)[0] < 1e-5
    lin1, lin2 = nn.Linear(128, 512).cuda(), nn.Linear(512, 128).cuda()
    x = _rand((2, 301, 128), 93)
    with torch.no_grad():
        h = tk.linear(lin1, x, act=3)
        y = tk.linear(lin2, h, res=x)
        ref_h = torch.nn.functional.gelu(torch.nn.functional.linear(x.double(), lin1.weight.double(), lin1.bias.double()))
        ref = x.double() + torch.nn.functional.linear(ref_h, lin2.weight.double(), lin2.bias.double())
    assert _relerr(h, ref_h)[0] < 2e-5 and _relerr(y, ref)[0] < 2e-5
    with pytest.raises(C.HipExtensionError):
        tk.linear(lin1, x.cpu())


@pytest.mark.parametrize("d", [32, 64])
def test_window_attn(C, d):
    """Part-head window attention kernel against the reference formulation (window_partition / nn.Unfold + softmax)."""
    import torch.nn.functional as F

    from iggt_official_amd.heads.window_sa import window_partition, window_reverse

    b, h, w, nh = 2, 16, 24, 4
    c = nh * d
    scale = d ** -0.5
    # (a) HAB: q, k, v slices of one [b,h,w,3c] map, 8x8 windows, no bias
    qkv = _rand((b, h, w, 3 * c), 100, 1.0)
    o = torch.full((b, h, w, c), float("nan"), device="cuda")
    C.window_attn(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], o, nh, d, scale)
    win = window_partition(qkv.double(), 8).view(-1, 64, 3, nh, d).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(win[0], win[1], win[2], scale=scale)
    ref = window_reverse(ref.transpose(1, 2).reshape(-1, 8, 8, c), 8, h, w)
    assert _relerr(o, ref)[0] < 1e-5
    # (b) OCAB: window-major queries, 12x12 zero-padded key/value windows, additive bias
    ow, pad = 12, 2
    nW = b * (h // 8) * (w // 8)
    qw = _rand((nW, 64, c), 101, 1.0)
    kk, vv = _rand((b, h, w, c), 102, 1.0), _rand((b, h, w, c), 103, 1.0)
    bias = _rand((nh, 64, ow * ow), 104, 0.5)                      # [head][query][key] as the reference builds it
    o = torch.full((b, h, w, c), float("nan"), device="cuda")
    C.window_attn(qw, kk, vv, o, nh, d, scale, q_windows=True, ow=ow, pad=pad,
                  bias=bias.permute(0, 2, 1).contiguous())
    kv = F.unfold(torch.cat((kk, vv), -1).permute(0, 3, 1, 2).double(), kernel_size=(ow, ow), stride=8, padding=pad)
    kv = kv.view(b, 2, c, ow * ow, -1).permute(1, 0, 4, 3, 2).reshape(2, nW, ow * ow, c)
    qh = qw.double().view(nW, 64, nh, d).transpose(1, 2)
    kh, vh = (t.view(nW, ow * ow, nh, d).transpose(1, 2) for t in (kv[0], kv[1]))
    ref = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=bias.double().unsqueeze(0), scale=scale)
    ref = window_reverse(ref.transpose(1, 2).reshape(-1, 8, 8, c), 8, h, w)
    assert _relerr(o, ref)[0] < 1e-5
