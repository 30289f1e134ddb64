"""The x3 precision rung (GPU; round 5, review item 1): fp16 hi + lo operand pairs, three MFMA passes per product.

Kernel level, through the C ABI (include/iggt_hip.h "x3 precision rung"; csrc/x3.hip): every producer of split operands against an
fp64 evaluation of the same op (what comes out must reconstruct, hi + lo, to 2^-21 of the value or fp16's subnormal grid), the
three-pass GEMM over the concatenated K axis against the fp64 product of the reconstructed operands, and the attention on pairs
against an fp64 softmax attention of the reconstructed q, k, v -- at ragged shapes, at the frame shape of the 518^2 configurations
and (row-sampled) at the global shape of 8 views.  Tolerances state what three fp16 passes can give: ~1e-6 of the result, against
the 2.4e-4 of a single fp16 rounding.

Block level: one escalated block against an fp64 restatement of the reference block (reference block.py:81-107, attention.py:50-77,
mlp.py:34-39) on heavy-tailed parameters: 1e-5, where the single-fp16 block is at 1e-3.
Model level (the doses the rung is for): tests/test_trained_like_gpu.py."""
import pytest
import torch

from conftest import report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def C():
    from iggt_official_amd import _C

    _C.load()
    return _C


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).cuda()


def _pair_err(hi, lo, ref):
    """max |hi + lo - ref| relative to max(|ref| 2^-21, fp16 subnormal step): <= 1 means "as good as an fp16 pair can be"."""
    got = hi.double() + lo.double()
    tol = torch.maximum(ref.double().abs() * 2.0 ** -21, torch.tensor(2.0 ** -24, dtype=torch.float64, device=ref.device))
    return float(((got - ref.double()).abs() / tol).max())


def test_layernorm_split3(C):
    T, Cc = 777, 1024
    x = _rand((T, Cc), 1, 3.0) + 0.5
    w = torch.exp(_rand((Cc,), 2))            # log-normal scales: the statistics the rung is for
    b = _rand((Cc,), 3, 0.1)
    out = torch.full((T, 3 * Cc), float("nan"), dtype=torch.float16, device="cuda")
    C.layernorm(x, w, b, out, 1e-5, split3=True)
    ref = torch.nn.functional.layer_norm(x.double(), (Cc,), w.double(), b.double(), 1e-5)
    assert torch.equal(out[:, :Cc], out[:, 2 * Cc:])                       # [hi | lo | hi]
    assert torch.equal(out[:, :Cc], ref.float().half()) or float((out[:, :Cc].double() - ref).abs().max()) < 0.51 * 2.0 ** -10 * float(ref.abs().max())
    # the pair reconstructs the fp32 LayerNorm result; that result itself is ~1e-6 from fp64
    ref32 = torch.nn.functional.layer_norm(x, (Cc,), w, b, 1e-5)
    e = float(((out[:, :Cc].double() + out[:, Cc:2 * Cc].double()) - ref).norm() / ref.norm())
    report("x3/layernorm_split3", dict(l2=e, fp32_l2=float((ref32.double() - ref).norm() / ref.norm())))
    assert e < 2e-6, e


def test_split3_and_gelu(C):
    rows, N = 301, 4096
    x = _rand((rows, N), 4, 2.0)
    for act in (0, 1):
        out = torch.full((rows, 3 * N), float("nan"), dtype=torch.float16, device="cuda")
        C.split3(x, out, act=act)
        ref = torch.nn.functional.gelu(x.double()) if act else x.double()
        assert torch.equal(out[:, :N], out[:, 2 * N:])
        got = out[:, :N].double() + out[:, N:2 * N].double()
        l2 = float((got - ref).norm() / ref.norm())
        report(f"x3/split3_act{act}", dict(l2=l2, pair=_pair_err(out[:, :N], out[:, N:2 * N], ref.float())))
        assert l2 < (3e-7 if act == 0 else 1e-6), (act, l2)      # act 1: the fp32 erf-GELU before the split


def test_im2row_split3(C):
    S, H, W, KPAD = 2, 28, 42, 640
    img = torch.rand(S, 3, H, W, generator=torch.Generator().manual_seed(60)).cuda()
    rows = S * (H // 14) * (W // 14)
    single = torch.empty(rows, KPAD, dtype=torch.float16, device="cuda")
    C.im2row_patch14(img, single, S, H, W, KPAD)
    out = torch.full((rows, 3 * KPAD), float("nan"), dtype=torch.float16, device="cuda")
    C.im2row_patch14(img, out, S, H, W, KPAD, split3=True)
    assert torch.equal(out[:, :KPAD], single) and torch.equal(out[:, 2 * KPAD:], single)
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float64, device="cuda").view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float64, device="cuda").view(1, 3, 1, 1)
    ref = torch.nn.functional.unfold((img.double() - mean) / std, 14, stride=14).transpose(1, 2).reshape(rows, 588)
    got = out[:, :588].double() + out[:, KPAD:KPAD + 588].double()
    assert float((got - ref).abs().max()) < 1e-6 * float(ref.abs().max())
    assert float(out[:, KPAD + 588:2 * KPAD].abs().max()) == 0.0


def test_gemm_over_concatenated_k(C):
    """A' = [A_hi | A_lo | A_hi] against W' = [W_hi | W_hi | W_lo]: one fp16 GEMM = the three-pass product."""
    from iggt_official_amd.layers.blocks import _x3_weight

    for M, N, K in ((1374, 3072, 1024), (300, 1024, 4096), (9000, 1024, 1024)):
        a = _rand((M, K), 5, 2.0)
        w = _rand((N, K), 6, K ** -0.5)
        a3 = torch.empty(M, 3 * K, dtype=torch.float16, device="cuda")
        C.split3(a, a3)
        w3 = _x3_weight(w)
        bias = _rand((N,), 7)
        out = torch.full((M, N), float("nan"), device="cuda")
        C.gemm_h16(a3, w3, out, bias=bias)
        ref = a.double() @ w.double().t() + bias.double()
        e_pair = float((out.double() - ref).norm() / ref.norm())
        single = torch.empty(M, N, device="cuda")
        C.gemm_h16(a.half(), w.half(), single, bias=bias)
        e_single = float((single.double() - ref).norm() / ref.norm())
        report(f"x3/gemm_{M}x{N}x{K}", dict(l2_pairs=e_pair, l2_single_fp16=e_single))
        assert e_pair < 2e-6 and e_pair < e_single / 50, (e_pair, e_single)
        # residual epilogue (LayerScale + accumulate), as proj / fc2 use it
        x = _rand((M, N), 8)
        x0 = x.clone()
        gamma = _rand((N,), 9) * 0.1 + 1
        C.gemm_h16(a3, w3, x, bias=bias, gamma=gamma, accumulate=True)
        ref2 = x0.double() + gamma.double() * ref
        assert float((x.double() - ref2).norm() / ref2.norm()) < 2e-6


def _rope_ref(t, pos, base=100.0):
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


@pytest.mark.parametrize("with_norm", [True, False])
def test_qkv_split(C, with_norm):
    from iggt_official_amd.layers.rope import RotaryPositionEmbedding2D

    S, gh, gw, psi = 2, 5, 7, 5
    P = psi + gh * gw
    T = S * P
    qkv = _rand((T, 3072), 50, 1.5)
    qw, qb, kw, kb = torch.exp(_rand((64,), 51)), _rand((64,), 52, 0.1), torch.exp(_rand((64,), 53)), _rand((64,), 54, 0.1)
    cos, sin = RotaryPositionEmbedding2D(100).tables(64, max(gh, gw), torch.device("cuda"))
    out = torch.full((T, 6 * 1024), float("nan"), dtype=torch.float16, device="cuda")
    scale = 0.125 * C.LOG2E
    kw_ = dict(qw=qw, qb=qb, kw=kw, kb=kb, cos_t=cos, sin_t=sin, P=P, gw=gw, patch_start=psi, eps=1e-5) if with_norm else {}
    C.qkv_split(qkv, out[:, :1024], 3072, out[:, 1024:2048], 3072, out[:, 2048:3072], 3072, q_scale=scale, **kw_)
    assert not torch.isnan(out.float()).any()
    pos = torch.zeros(P, 2, dtype=torch.long)
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    pos[psi:, 0], pos[psi:, 1] = ys.flatten() + 1, xs.flatten() + 1
    pos = pos.repeat(S, 1)
    for idx, (w_, b_, sc) in enumerate([(qw, qb, scale), (kw, kb, 1.0), (None, None, 1.0)]):
        t = qkv[:, idx * 1024:(idx + 1) * 1024].double().cpu()
        if with_norm and idx < 2:
            t = torch.nn.functional.layer_norm(t.view(T, 16, 64), (64,), w_.double().cpu(), b_.double().cpu(), 1e-5)
            t = _rope_ref(t, pos).reshape(T, 1024)
        ref = t * sc
        got = (out[:, idx * 1024:(idx + 1) * 1024].double() + out[:, 3072 + idx * 1024:3072 + (idx + 1) * 1024].double()).cpu()
        l2 = float((got - ref).norm() / ref.norm())
        report(f"x3/qkv_split_norm{int(with_norm)}_{'qkv'[idx]}", dict(l2=l2))
        assert l2 < 2e-6, (idx, l2)          # fp32 LayerNorm + RoPE arithmetic, then a 22-bit pair


def _attn_ref(q, k, v, rows=None):
    """fp64 softmax(q k^T) v; q already carries scale * log2 e (base-2 softmax).  q [B, H, Nq, 64] etc."""
    if rows is not None:
        q = q[:, :, rows]
    s = q.double() @ k.double().transpose(-1, -2)
    p = torch.exp2(s - s.amax(-1, keepdim=True))
    return (p @ v.double()) / p.sum(-1, keepdim=True)


@pytest.mark.parametrize("B,H,Nq,Nk,sharp", [(2, 16, 300, 300, 1.0), (1, 16, 77, 1000, 4.0), (3, 16, 1374, 1374, 1.0),
                                               (1, 16, 10992, 10992, 6.0)])
def test_flash_attn_x3(C, B, H, Nq, Nk, sharp):
    """Token-major [B][N][H * 64] pairs; `sharp` scales q: logits of std ~ sharp * 1.4 * ... (near-argmax rows at 6)."""
    Cc = H * 64
    src = _rand((3, B, max(Nq, Nk), Cc), 70 + Nq % 7, 1.0)
    src[0] *= sharp * 0.125 * C.LOG2E * 3.0
    buf = torch.empty(B * max(Nq, Nk), 6 * Cc, dtype=torch.float16, device="cuda")    # [q_hi k_hi v_hi q_lo k_lo v_lo]
    for i in range(3):
        x = src[i].reshape(-1, Cc)
        hi = x.half()
        buf[:, i * Cc:(i + 1) * Cc] = hi
        buf[:, (3 + i) * Cc:(4 + i) * Cc] = (x - hi.float()).half()
    N = max(Nq, Nk)
    o = torch.full((B * N, 3 * Cc), float("nan"), dtype=torch.float16, device="cuda")
    C.flash_attn_x3(buf, buf[:, 3 * Cc:], buf[:, Cc:], buf[:, 4 * Cc:], buf[:, 2 * Cc:], buf[:, 5 * Cc:], o, Cc, B, H, Nq, Nk,
                    N * 6 * Cc, 6 * Cc, N * 6 * Cc, 6 * Cc, N * 6 * Cc, 6 * Cc, N * 3 * Cc, 3 * Cc)
    torch.cuda.synchronize()
    rec = (buf[:, :3 * Cc].double() + buf[:, 3 * Cc:].double()).view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)   # [3, B, H, N, 64]
    rows = torch.arange(0, Nq, max(1, Nq // 512), device="cuda") if Nq > 2000 else torch.arange(Nq, device="cuda")
    ref = _attn_ref(rec[0][:, :, :Nq], rec[1][:, :, :Nk], rec[2][:, :, :Nk], rows)      # [B, H, rows, 64]
    ov = o.view(B, N, 3, H, 64)
    assert torch.equal(ov[:, :Nq, 0], ov[:, :Nq, 2])
    got = (ov[:, :, 0].double() + ov[:, :, 1].double())[:, rows].permute(0, 2, 1, 3)
    l2 = float((got - ref).norm() / ref.norm())
    mx = float((got - ref).abs().max() / ref.abs().max())
    # the same attention on single fp16 operands (the hi parts), for scale
    single = torch.empty(B * N, Cc, dtype=torch.float16, device="cuda")
    C.flash_attn_d64(buf, buf[:, Cc:], buf[:, 2 * Cc:], single, B, H, Nq, Nk, N * 6 * Cc, 6 * Cc, N * 6 * Cc, 6 * Cc, N * 6 * Cc,
                     6 * Cc, N * Cc, Cc, 1.0 / C.LOG2E)
    l2_single = float((single.view(B, N, H, 64).double()[:, rows].permute(0, 2, 1, 3) - ref).norm() / ref.norm())
    report(f"x3/attn_B{B}_Nq{Nq}_Nk{Nk}_sharp{sharp}", dict(l2=l2, max=mx, l2_single_fp16=l2_single))
    assert l2 < 3e-6 and mx < 1e-5, (l2, mx, l2_single)
    if Nq < N:   # rows past Nq are not written
        assert torch.isnan(ov[:, Nq:].float()).all()


def test_block_x3_against_fp64():
    """One q/k-norm + RoPE block with log-normal (sigma = 1) norm scales, escalated, against an fp64 restatement of the reference
    block; the same block on single fp16 operands for scale."""
    from iggt_official_amd import precision
    from iggt_official_amd.layers.blocks import Block, Workspace
    from iggt_official_amd.layers.rope import RotaryPositionEmbedding2D
    from oracle import restate

    torch.manual_seed(0)
    S, gh, gw, psi = 3, 6, 5, 5
    P = psi + gh * gw
    T = S * P
    rope = RotaryPositionEmbedding2D(100)
    blk = Block(dim=1024, num_heads=16, qk_norm=True, init_values=1.0, rope=rope).cuda().eval()
    with torch.no_grad():
        for n, p_ in blk.named_parameters():
            if n.endswith("norm1.weight") or n.endswith("norm2.weight") or "q_norm.weight" in n or "k_norm.weight" in n:
                p_.copy_(torch.exp(torch.randn_like(p_)))
            elif n.endswith(".bias"):
                p_.copy_(torch.randn_like(p_) * 0.1)
            elif n.endswith("gamma"):
                p_.copy_(torch.rand_like(p_) + 0.5)
    x = torch.randn(T, 1024, device="cuda")
    cos, sin = rope.tables(64, max(gh, gw), torch.device("cuda"))
    geom = dict(P=P, gw=gw, patch_start=psi, cos=cos, sin=sin)
    sd = {"b." + k: v.detach().double().cpu() for k, v in blk.state_dict().items()}
    pos = torch.zeros(P, 2, dtype=torch.long)
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    pos[psi:, 0], pos[psi:, 1] = ys.flatten() + 1, xs.flatten() + 1
    ref = restate.block(sd, "b", x.double().cpu().view(S, P, 1024), 16, pos[None].expand(S, P, 2)).reshape(T, 1024)
    res = {}
    for mode in ("all", "off"):
        precision.set_escalation(mode)
        try:
            y = x.clone()
            blk.forward_inplace(y, Workspace(), batch=S, tokens=P, rope_geom=geom)
            assert blk.packed()["x3"] == (mode == "all")
            res[mode] = float((y.double().cpu() - ref).norm() / ref.norm())
        finally:
            precision.set_escalation("auto")
    report("x3/block_vs_fp64", dict(l2_x3=res["all"], l2_single_fp16=res["off"]))
    assert res["all"] < 2e-5 and res["all"] < res["off"] / 20, res


def test_gemm_row_chunks_equal_one_launch(C):
    """layers/blocks.py _gemm_rows: above the 32-bit operand-offset limit of the LDS-DMA GEMMs (the K = 12 288 GEMM of the rung at
    64 views @ 1036^2) the rows go in chunks; here the limit is lowered so that a small problem takes the chunked path."""
    from iggt_official_amd.layers import blocks

    M, N, K = 5000, 1024, 768
    a = _rand((M, K), 11).half()
    w = _rand((N, K), 12, K ** -0.5).half()
    bias, gamma = _rand((N,), 13), _rand((N,), 14) * 0.1 + 1
    x0 = _rand((M, N), 15)
    one = x0.clone()
    C.gemm_h16(a, w, one, bias=bias, gamma=gamma, accumulate=True)
    old = blocks.GEMM_MAX_OPERAND_ELEMENTS
    blocks.GEMM_MAX_OPERAND_ELEMENTS = 1 << 20          # 1 024 rows of 768 per chunk (a multiple of 256)
    try:
        many = x0.clone()
        blocks._gemm_rows(a, w, many, bias=bias, gamma=gamma, accumulate=True)
    finally:
        blocks.GEMM_MAX_OPERAND_ELEMENTS = old
    ref = x0.double() + gamma.double() * (a.double() @ w.double().t() + bias.double())
    assert float((many.double() - ref).norm() / ref.norm()) < 2e-6
    assert float((many - one).abs().max()) < 1e-5 * float(one.abs().max())     # different tile kernels may serve the two forms


@pytest.mark.parametrize("case", ["tiny_s2_56_stress", "tiny_s3_84x56_stress"])
def test_whole_model_on_the_rung_at_small_shapes(case):
    """Every block forced onto the rung (IGGT_ESCALATE=all) on the bounded-uniform checkpoint at the smallest shapes -- 21 / 29
    tokens per view: the GEMMs fall to the 128^2 kernel with K = 3 072 / 12 288, the pair attention runs ragged tiles, the part
    head consumes the tokens -- and as a batch of two scenes (B = 2).  The aggregated tokens must sit an order of magnitude closer
    to the reference than on single operands; every output inside its usual gate."""
    from conftest import load_golden
    from helpers import build_gpu_model, errors
    from iggt_official_amd import precision
    from oracle import weights

    g = load_golden(case)
    m = g["meta"]
    model = build_gpu_model(m["mode"], m["weight_seed"])
    images = weights.make_images(m["S"], m["H"], m["W"], seed=m["image_seed"], device="cuda")
    cap = {}
    hook = model.aggregator.register_forward_hook(lambda mod, i, o: cap.__setitem__("tokens", o[0]))
    res = {}
    try:
        for mode in ("off", "all"):
            precision.set_escalation(mode)
            pred = model(images)
            torch.cuda.synchronize()
            assert len(model.aggregator.escalation_report()["x3"]) == (72 if mode == "all" else 0)
            res[mode] = {f"tokens_{li}": errors(cap["tokens"][li], g[f"tokens_{li}"])[1] for li in (4, 11, 17, 23)}
            for k in ("depth", "world_points", "part_feat"):
                if k in g:
                    res[mode][k] = errors(pred[k], g[k])[1]
            res[mode]["pose_enc"] = errors(torch.stack(pred["pose_enc"], 0), g["pose_enc"])[1]
        two = model(torch.stack([images, images.flip(0)], 0))        # B = 2 scenes on the rung
        torch.cuda.synchronize()
        assert two["depth"].shape[0] == 2 and errors(two["depth"][:1], pred["depth"])[1] < 1e-6
    finally:
        precision.set_escalation("auto")
        hook.remove()
    report(f"x3/whole_model/{case}", res)
    for k, v in res["all"].items():
        assert v < 1e-3, (k, v)
    assert res["all"]["tokens_23"] < res["off"]["tokens_23"] / 10, res
