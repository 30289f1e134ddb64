"""fp16-operand twins of the trunk kernels (include/iggt_hip.h `_f16` entry points), the default operand format of
the host model (iggt_official_amd/precision.py).  Same structure as test_kernels_gpu.py: fp64 reference on the SAME
fp16-rounded operands; admissible differences are accumulation order and the documented internal roundings, now at
fp16's 2^-11.  Tolerances are stated per test (8x tighter than the bf16 ones where a 16-bit rounding is involved)."""
import pytest
import torch

from conftest import report
from test_kernels_gpu import _attn_ref, _rand, _relerr, _rope_ref

pytestmark = pytest.mark.gpu
F16 = torch.float16


@pytest.fixture(scope="module")
def C():
    from iggt_official_amd import _C

    _C.load()
    return _C


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1374, 3072, 1024), (77, 1024, 4096), (1000, 96, 640), (33, 9, 128)])
def test_gemm_f16_plain_f32_out(C, M, N, K):
    a = _rand((M, K), 1, dtype=F16)
    w = _rand((N, K), 2, K ** -0.5, dtype=F16)
    bias = _rand((N,), 3)
    out = torch.full((M, N), float("nan"), device="cuda")
    C.gemm_h16(a, w, out, bias=bias)
    ref = a.double() @ w.double().t() + bias.double()
    mx, l2 = _relerr(out, ref)
    report(f"gemm_f16_f32out_{M}x{N}x{K}", dict(max=mx, l2=l2))
    assert mx < 2e-5, (mx, l2)  # fp32 accumulation of exact fp16 products


def test_gemm_f16_mixed_operands_rejected(C):
    a = _rand((128, 64), 1, dtype=F16)
    w = _rand((128, 64), 2, dtype=torch.bfloat16)
    with pytest.raises(C.HipExtensionError):
        C.gemm_h16(a, w, torch.empty(128, 128, device="cuda"))


@pytest.mark.parametrize("M,N,K,mode", [(5000, 4096, 1024, "gelu"), (4122, 1024, 4096, "acc"), (2738, 1024, 640, "remap"),
                                        (43968, 3072, 1024, "plain"), (300, 256, 128, "gelu"),
                                        (9000, 4096, 1024, "gelu"), (8448, 1024, 4096, "acc"), (5496, 3072, 1024, "gelu"),
                                        (5496, 4096, 1024, "gelu"), (5496, 1024, 4096, "acc"), (5400, 1024, 4096, "plain")])
def test_gemm_f16_epilogues(C, M, N, K, mode):
    """All three GEMM kernels: 1 024 <= M < 8 192 is the 256 x 128 two-workgroups-per-CU kernel (gemm_bf16_duo.hip: the
    per-rank shapes of a sharded run), M >= 8 192 the 256 x 256 LDS-DMA ping-pong kernel (gemm_bf16_t256.hip: ragged last
    row tile, GELU from the LDS table / LayerScale-accumulate epilogues), the small shape the 128 x 128 kernel; M = 5 496 x N =
    3 072 | 4 096 | 1 024 (per-rank qkv / fc1 / fc2) and the fp32-output case at M = 5 400 take the 192-row variant of the
    two-workgroups-per-CU kernel (fewer rounds x rows of the 512 workgroup slots; the last row tile is 120 | 24 rows)."""
    a = _rand((M, K), 70, dtype=F16)
    w = _rand((N, K), 71, K ** -0.5, dtype=F16)
    bias, gamma = _rand((N,), 72, 0.1), _rand((N,), 73)
    base = (a.double() @ w.double().t() + bias.double()) if M < 20000 else None
    if mode == "gelu":
        out = torch.empty(M, N, dtype=F16, device="cuda")
        C.gemm_h16(a, w, out, bias=bias, act=1)
        mx, l2 = _relerr(out, torch.nn.functional.gelu(base))
        report(f"gemm_f16_gelu_{M}", dict(max=mx, l2=l2))
        assert mx < 8e-4 and l2 < 4e-4  # one fp16 rounding of the output (2^-11 relative) + A&S erfc (1.5e-7) / table (2.5e-6)
    elif mode == "acc":
        x = _rand((M, N), 74)
        ref = x.double() + gamma.double() * base
        C.gemm_h16(a, w, x, bias=bias, gamma=gamma, accumulate=True)
        assert _relerr(x, ref)[0] < 2e-5
    elif mode == "remap":
        g2 = 1369
        S = M // g2
        table = _rand((g2, N), 75)
        out = torch.full((S * (g2 + 5), N), -7.0, device="cuda")
        C.gemm_h16(a, w, out, bias=bias, add_table=table, rows_in=g2, rows_out=g2 + 5, row_off=5)
        o = out.view(S, g2 + 5, N)
        assert torch.all(o[:, :5] == -7.0)
        assert _relerr(o[:, 5:], base.view(S, g2, N) + table.double())[0] < 2e-5
    else:
        out = torch.full((M, N), float("nan"), device="cuda")
        C.gemm_h16(a, w, out, bias=bias)
        rows = torch.arange(0, M, 97, device="cuda")
        base_s = a[rows].double() @ w.double().t() + bias.double()
        assert not torch.isnan(out).any() and _relerr(out[rows], base_s)[0] < 2e-5


@pytest.mark.parametrize("M,N", [(9216, 1024), (5496, 4096)])
def test_gemm_gelu_table_against_erf(C, M, N):
    """The fc1 epilogue of the 256 x 256 kernel evaluates GELU from an LDS table of Phi (gemm_common.h gelu_lut); the second shape
    is the per-rank fc1 on the 192-row two-workgroups-per-CU kernel (polynomial erfc), held to the same bound.  A GEMM whose result IS a chosen
    pre-activation (one non-zero operand column, unit weights) sweeps x over [-10, 10] and beyond the table; against torch's erf
    GELU (reference iggt/layers/mlp.py:34, nn.GELU()) in fp64 the error must stay within one fp16 rounding of the result plus
    the table's 2.5e-6."""
    K = 128
    xs = torch.cat([torch.linspace(-10, 10, M - 16, device="cuda"),
                    torch.tensor([-65000., -100., -8.0, -7.99, -1e-4, 0., 1e-4, 7.99, 8.0, 8.01, 100., 3000., 65000., 0.5, -0.5, 1.0],
                                 device="cuda")])
    a = torch.zeros(M, K, dtype=F16, device="cuda")
    a[:, 0] = xs.to(F16)
    w = torch.zeros(N, K, dtype=F16, device="cuda")
    w[:, 0] = 1.0
    out = torch.empty(M, N, dtype=F16, device="cuda")
    C.gemm_h16(a, w, out, act=1)
    x = a[:, 0].double()
    ref = torch.nn.functional.gelu(x).clamp(-65504, 65504)
    got = out.double()
    assert torch.equal(out[:, :1].expand(-1, N), out)                 # every column is the same function of x
    err = (got[:, 0] - ref).abs()
    tol = ref.abs() * 2.0 ** -11 + 3e-6
    assert bool((err <= tol).all()), (float((err - tol).max()), float(x[(err - tol).argmax()]))
    report(f"gemm_f16_gelu_table/M{M}", dict(max_abs=float(err.max()), max_abs_at=float(x[err.argmax()])))


def test_gemm_f16_store_saturates(C):
    """fp16 outputs beyond the finite range are clamped to +-65504 instead of becoming inf."""
    M, N, K = 256, 256, 128
    a = torch.full((M, K), 60.0, dtype=F16, device="cuda")
    w = torch.full((N, K), 60.0, dtype=F16, device="cuda")
    w[N // 2:] = -60.0
    out = torch.empty(M, N, dtype=F16, device="cuda")
    C.gemm_h16(a, w, out)   # 128 * 3600 = 460800 > 65504
    assert torch.all(out[:, : N // 2] == 65504.0) and torch.all(out[:, N // 2:] == -65504.0)


@pytest.mark.parametrize("B,H,Nq,Nk,tile", [(2, 16, 1374, 1374, 0), (1, 16, 4122, 4122, 0), (3, 4, 21, 21, 0),
                                            (1, 16, 4122, 4122, 5256), (1, 16, 2748, 5496, 6256),
                                            (2, 3, 1374, 1374, 6128), (1, 2, 300, 777, 6256), (1, 2, 300, 777, 5128),
                                            (1, 1, 40, 64, 6256), (1, 2, 1000, 65, 6256), (1, 2, 500, 129, 6128)])
def test_flash_attn_f16(C, B, H, Nq, Nk, tile):
    Cdim = H * 64
    N = max(Nq, Nk)
    qkv = _rand((B * N, 3 * Cdim), 20 + Nq, 1.0, F16)
    o = torch.full((B * Nq, Cdim), float("nan"), dtype=F16, device="cuda")
    C.flash_attn_d64(qkv, qkv[:, Cdim:], qkv[:, 2 * Cdim:], o, B, H, Nq, Nk,
                     N * 3 * Cdim, 3 * Cdim, N * 3 * Cdim, 3 * Cdim, N * 3 * Cdim, 3 * Cdim, Nq * Cdim, Cdim,
                     0.125, tile)
    x = qkv.view(B, N, 3, H, 64)
    q, k, v = x[:, :Nq, 0].transpose(1, 2), x[:, :Nk, 1].transpose(1, 2), x[:, :Nk, 2].transpose(1, 2)
    ref = _attn_ref(q, k, v, 0.125).transpose(1, 2).reshape(B * Nq, Cdim)
    assert not torch.isnan(o.float()).any()
    mx, l2 = _relerr(o, ref)
    report(f"attn_f16_B{B}_H{H}_{Nq}x{Nk}_t{tile}", dict(max=mx, l2=l2))
    # P and O are rounded to fp16 (2^-11 each): 8x below the bf16 kernel's 1.5e-2 / 4e-3
    assert mx < 2e-3 and l2 < 5e-4, (mx, l2)


def test_flash_attn_f16_peaked_and_wide_range(C):
    """(a) a dominant key in a late tile (online-softmax rescale); (b) one key 14 nats above a large diffuse crowd:
    the crowd's numerators sit near fp16's normal limit and still carry ~40 % of the mass -- the 2^P_SHIFT scaling of
    the numerators (csrc/attention_common.h) keeps them out of the subnormal range."""
    H, N = 2, 4000
    Cdim = H * 64
    qkv = _rand((N, 3 * Cdim), 33, 0.5, F16)
    x = qkv.view(N, 3, H, 64)
    x[700, 1] = x[123, 0] * 8.0
    x[901, 1] = x[5, 0] * 6.0
    # (b): query 77 of head 0: all keys orthogonal-ish (score ~0) except key 3000 with score ~ +8 nats
    x[77, 0, 0] = 0
    x[77, 0, 0, 0] = 8.0
    x[:, 1, 0, 0] = 0
    x[3000, 1, 0, 0] = 8.0            # score 64 * 0.125 = 8 nats above the crowd (crowd mass 3999 vs e^8 = 2981)
    o = torch.empty(N, Cdim, dtype=F16, device="cuda")
    C.flash_attn_d64(qkv, qkv[:, Cdim:], qkv[:, 2 * Cdim:], o, 1, H, N, N, 0, 3 * Cdim, 0, 3 * Cdim, 0, 3 * Cdim,
                     0, Cdim, 0.125, 0)
    q, k, v = x[:, 0].transpose(0, 1), x[:, 1].transpose(0, 1), x[:, 2].transpose(0, 1)
    ref = _attn_ref(q, k, v, 0.125).transpose(0, 1).reshape(N, Cdim)
    mx, l2 = _relerr(o, ref)
    assert mx < 2e-3 and l2 < 5e-4, (mx, l2)
    row = _relerr(o[77, :64], ref[77, :64])
    assert row[1] < 1e-3, row


def test_flash_attn_unknown_tile_code_rejected(C):
    """Only the production kernel's tile codes exist (the round-1 A/B kernels 128 / 256 / 512 are no longer built)."""
    qkv = _rand((64, 3 * 64), 1, 1.0, F16)
    o = torch.empty(64, 64, dtype=F16, device="cuda")
    with pytest.raises(C.HipExtensionError):
        C.flash_attn_d64(qkv, qkv[:, 64:], qkv[:, 128:], o, 1, 1, 64, 64, 0, 192, 0, 192, 0, 192, 0, 64, 0.125, 128)


# ---------------------------------------------------------------------------------------------
# Static-bound softmax (csrc/attention_v3.hip, iggt_flash_attn_static_*): q carries scale * log2(e), the shift is
# |q|max |k|max per head.  The reference is the same fp64 softmax (base 2 on the pre-scaled q = scale ln 2).
def _static_attn(C, qkv, B, H, Nq, Nk, N, tile, dtype):
    Cdim = H * 64
    x = qkv.view(B, N, 3, H, 64)
    qn = x[:, :Nq, 0].float().norm(dim=-1).amax(dim=(0, 1))      # [H]
    kn = x[:, :Nk, 1].float().norm(dim=-1).amax(dim=(0, 1))
    qkmax = torch.zeros(32, device="cuda")
    qkmax[:H], qkmax[16:16 + H] = qn, kn
    rows = 256 if tile in (5256, 6256) else 128          # tile 0 (auto) may pick either: size for the finer one
    nwork = B * H * ((Nq + rows - 1) // rows) if tile else B * H * ((Nq + 127) // 128)
    flags = torch.full((nwork + 3,), 7, dtype=torch.int32, device="cuda")
    o = torch.full((B * Nq, Cdim), float("nan"), dtype=dtype, device="cuda")
    C.flash_attn_d64_static(qkv, qkv[:, Cdim:], qkv[:, 2 * Cdim:], o, B, H, Nq, Nk,
                            N * 3 * Cdim, 3 * Cdim, N * 3 * Cdim, 3 * Cdim, N * 3 * Cdim, 3 * Cdim, Nq * Cdim, Cdim,
                            qkmax, flags, tile)
    q, k, v = x[:, :Nq, 0].transpose(1, 2), x[:, :Nk, 1].transpose(1, 2), x[:, :Nk, 2].transpose(1, 2)
    ref = _attn_ref(q, k, v, 0.6931471805599453).transpose(1, 2).reshape(B * Nq, Cdim)
    return o, ref, flags


@pytest.mark.parametrize("dtype", [F16, torch.bfloat16])
@pytest.mark.parametrize("normed", [True, False])
@pytest.mark.parametrize("B,H,Nq,Nk,tile", [(2, 16, 1374, 1374, 0), (1, 16, 4122, 4122, 0), (3, 4, 21, 21, 0),
                                            (1, 16, 4122, 4122, 5256), (1, 16, 2748, 5496, 6256),
                                            (2, 3, 1374, 1374, 6128), (1, 2, 300, 777, 6256), (1, 2, 300, 777, 5128),
                                            (1, 1, 40, 64, 6256), (1, 2, 1000, 65, 6256), (1, 2, 500, 129, 6128)])
def test_flash_attn_static_bound(C, dtype, normed, B, H, Nq, Nk, tile):
    """normed: q and k head vectors of equal norm, as the model's per-head LayerNorm makes them (|.| = 8): the bound is
    tight and no tile may need the fallback.  Not normed: Gaussian vectors whose largest norms sit ~35 % above the typical
    one -- the bound loosens by several bits, some tiles take the fallback pass; the result must be right either way."""
    Cdim = H * 64
    N = max(Nq, Nk)
    qkv = _rand((B * N, 3 * Cdim), 120 + Nq, 1.0, dtype)
    if normed:
        qk = qkv[:, :2 * Cdim].float().view(B * N, 2 * H, 64)
        qkv[:, :2 * Cdim] = (qk / qk.norm(dim=-1, keepdim=True) * 8.0).view(B * N, 2 * Cdim).to(dtype)
    qkv[:, :Cdim] *= 0.125 * 1.4426950408889634 * 1.3      # "pre-scaled" q: scores ~ N(0, (1.9 bit)^2)
    o, ref, flags = _static_attn(C, qkv, B, H, Nq, Nk, N, tile, dtype)
    assert not torch.isnan(o.float()).any()
    mx, l2 = _relerr(o, ref)
    name = "f16" if dtype == F16 else "bf16"
    report(f"attn_static_{name}_B{B}_H{H}_{Nq}x{Nk}_t{tile}", dict(max=mx, l2=l2))
    if dtype == F16:
        assert mx < 2e-3 and l2 < 5e-4, (mx, l2)
    else:
        assert mx < 1.5e-2 and l2 < 4e-3, (mx, l2)
    assert torch.all(flags[-3:] == 7)                      # scratch beyond the work list untouched
    if tile and normed:
        assert int(flags[:-3].sum()) == 0                  # LayerNorm-like operands: nothing needs the fallback


def test_flash_attn_static_bound_fallback_rows(C):
    """Rows the static bound cannot serve in fp16 -- every score far below |q|max |k|max -- are flagged by their row sum and
    recomputed by the online-max kernel: (a) query rows with a tiny norm next to rows with a large one (bound 40 bits,
    their scores ~0), (b) a one-hot row, (c) ordinary rows.  All must meet the kernel tolerance."""
    H, N = 2, 4000
    Cdim = H * 64
    qkv = _rand((N, 3 * Cdim), 133, 1.0, F16)
    x = qkv.view(N, 3, H, 64)
    x[:, 0] *= 0.125 * 1.4426950408889634
    x[:, 0] *= 3.0                                 # |q^| ~ 4.3, |k| ~ 8  ->  bound ~ 35 bits per head
    x[::3, 0] *= 0.01                              # every third query row: scores within +-0.1 bit, 35 bits below the bound
    x[701, 1] = x[124, 0] / x[124, 0].float().norm(dim=-1, keepdim=True).half() * 8.0   # key 701 aligned with query 124
    o, ref, flags = _static_attn(C, qkv, 1, H, N, N, N, 0, F16)
    assert not torch.isnan(o.float()).any()
    mx, l2 = _relerr(o, ref)
    nflag = int((flags[:-3] == 1).sum())
    report("attn_static_f16_fallback", dict(max=mx, l2=l2, flagged=nflag, tiles=int(flags.numel() - 3)))
    assert nflag > 0                               # the fallback really ran
    assert mx < 2e-3 and l2 < 5e-4, (mx, l2)
    for r in (0, 3, 124, 125, 3999):
        e = _relerr(o[r], ref[r])
        assert e[1] < 1e-3, (r, e)


@pytest.mark.parametrize("dtype", [F16, torch.bfloat16])
@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 2, 300, 4000), (1, 16, 1374, 5496), (2, 3, 500, 2100)])
def test_flash_attn_static_key_split(C, dtype, B, H, Nq, Nk):
    """Small grids: with a partial workspace the dispatcher splits the keys into ranges (partial results add under the common
    static bound) -- the same answer as one pass; and the two-segment form the multi-GPU path uses (own keys first, the
    other ranks' keys later, explicit slots) gives it too."""
    Cdim = H * 64
    N = max(Nq, Nk)
    qkv = _rand((B * N, 3 * Cdim), 170 + Nq, 1.0, dtype)
    qk = qkv[:, :2 * Cdim].float().view(B * N, 2 * H, 64)
    qkv[:, :2 * Cdim] = (qk / qk.norm(dim=-1, keepdim=True) * 8.0).view(B * N, 2 * Cdim).to(dtype)
    qkv[:, :Cdim] *= 0.125 * 1.4426950408889634 * 1.3
    x = qkv.view(B, N, 3, H, 64)
    qkmax = torch.zeros(32, device="cuda")
    qkmax[:H] = x[:, :Nq, 0].float().norm(dim=-1).amax(dim=(0, 1))
    qkmax[16:16 + H] = x[:, :Nk, 1].float().norm(dim=-1).amax(dim=(0, 1))
    strides = (N * 3 * Cdim, 3 * Cdim, N * 3 * Cdim, 3 * Cdim, N * 3 * Cdim, 3 * Cdim)
    q, k, v = x[:, :Nq, 0].transpose(1, 2), x[:, :Nk, 1].transpose(1, 2), x[:, :Nk, 2].transpose(1, 2)
    ref = _attn_ref(q, k, v, 0.6931471805599453).transpose(1, 2).reshape(B * Nq, Cdim)
    tol = (2e-3, 5e-4) if dtype == F16 else (1.5e-2, 4e-3)
    flags = torch.zeros(B * H * ((Nq + 127) // 128), dtype=torch.int32, device="cuda")
    # (a) automatic split
    nws = C.static_attn_ws_bytes(B, H, Nq, Nk)
    assert nws > 0
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    o = torch.full((B * Nq, Cdim), float("nan"), dtype=dtype, device="cuda")
    C.flash_attn_d64_static(qkv, qkv[:, Cdim:], qkv[:, 2 * Cdim:], o, B, H, Nq, Nk, *strides, Nq * Cdim, Cdim, qkmax, flags,
                            0, ws)
    mx, l2 = _relerr(o, ref)
    assert not torch.isnan(o.float()).any() and mx < tol[0] and l2 < tol[1], (mx, l2)
    assert int(flags.sum()) == 0
    # (b) explicit segments: keys [0, n1) in one range, keys [n1, Nk) in three
    n1 = (Nk // 3 // 8) * 8
    o_part = torch.full((4, B, Nq, Cdim), float("nan"), dtype=dtype, device="cuda")
    l_part = torch.full((4, B, H, Nq), float("nan"), device="cuda")
    c_part = torch.full((4, B, H, Nq), float("nan"), device="cuda")
    kk, vv = qkv[:, Cdim:], qkv[:, 2 * Cdim:]
    C.flash_attn_d64_static_partial(qkv, kk, vv, B, H, Nq, n1, *strides, qkmax, o_part, l_part, c_part, 0, 1)
    C.flash_attn_d64_static_partial(qkv, kk[n1:], vv[n1:], B, H, Nq, Nk - n1, *strides, qkmax, o_part, l_part, c_part, 1, 3)
    o2 = torch.full((B * Nq, Cdim), float("nan"), dtype=dtype, device="cuda")
    C.flash_attn_d64_static_combine(o_part, l_part, c_part, 4, qkv, kk, vv, o2, B, H, Nq, Nk, *strides, Nq * Cdim, Cdim, flags)
    mx, l2 = _relerr(o2, ref)
    report(f"attn_static_split_{'f16' if dtype == F16 else 'bf16'}_B{B}_H{H}_{Nq}x{Nk}", dict(max=mx, l2=l2))
    assert not torch.isnan(o2.float()).any() and mx < tol[0] and l2 < tol[1], (mx, l2)
    # (c) the segments computed under DIFFERENT key bounds (multi-GPU: own keys under this rank's measured maximum, the
    # gathered keys under the maximum over all ranks): the combine step re-weights by the recorded shifts
    loose = qkmax.clone()
    loose[16:] *= 1.7
    o_part.fill_(float("nan")); l_part.fill_(float("nan")); c_part.fill_(float("nan"))
    C.flash_attn_d64_static_partial(qkv, kk, vv, B, H, Nq, n1, *strides, qkmax, o_part, l_part, c_part, 0, 1)
    C.flash_attn_d64_static_partial(qkv, kk[n1:], vv[n1:], B, H, Nq, Nk - n1, *strides, loose, o_part, l_part, c_part, 1, 3)
    assert float((c_part[1] - c_part[0]).min()) > 0.5          # the second segment really ran under a larger shift
    o3 = torch.full((B * Nq, Cdim), float("nan"), dtype=dtype, device="cuda")
    C.flash_attn_d64_static_combine(o_part, l_part, c_part, 4, qkv, kk, vv, o3, B, H, Nq, Nk, *strides, Nq * Cdim, Cdim, flags)
    mx, l2 = _relerr(o3, ref)
    assert not torch.isnan(o3.float()).any() and mx < tol[0] and l2 < tol[1], (mx, l2)
    # (d) segment mode, as a rank of a view-sharded run launches it: the keys are W "rank segments" of seg rows (the last one
    # ragged); this rank's own segment (index 1) goes first from a private copy, cut into 2 ranges -> slots W-1, W; then ONE
    # launch over the whole buffer with the own segment left out -> slots 0 .. W-2
    seg = ((Nk // 3 + 7) // 8) * 8 + 8
    Wn = (Nk + seg - 1) // seg
    assert Wn == 3 and Nk - 2 * seg > 0
    nsl = Wn - 1 + 2
    o_p = torch.full((nsl, B, Nq, Cdim), float("nan"), dtype=dtype, device="cuda")
    l_p = torch.full((nsl, B, H, Nq), float("nan"), device="cuda")
    c_p = torch.full((nsl, B, H, Nq), float("nan"), device="cuda")
    own_k, own_v = kk[seg:], vv[seg:]
    if B == 1:        # a private copy with its own row stride, like kv_local
        own = qkv[seg:2 * seg, Cdim:].clone()
        C.flash_attn_d64_static_partial(qkv, own, own[:, Cdim:], B, H, Nq, seg, strides[0], strides[1], 0, 2 * Cdim, 0, 2 * Cdim,
                                        qkmax, o_p, l_p, c_p, Wn - 1, 2)
    else:
        C.flash_attn_d64_static_partial(qkv, own_k, own_v, B, H, Nq, seg, *strides, qkmax, o_p, l_p, c_p, Wn - 1, 2)
    C.flash_attn_d64_static_partial(qkv, kk, vv, B, H, Nq, Nk, *strides, loose, o_p, l_p, c_p, 0, Wn, seg_len=seg, skip_seg=1)
    o4 = torch.full((B * Nq, Cdim), float("nan"), dtype=dtype, device="cuda")
    C.flash_attn_d64_static_combine(o_p, l_p, c_p, nsl, qkv, kk, vv, o4, B, H, Nq, Nk, *strides, Nq * Cdim, Cdim, flags)
    mx, l2 = _relerr(o4, ref)
    assert not torch.isnan(o4.float()).any() and mx < tol[0] and l2 < tol[1], (mx, l2)
    with pytest.raises(C.HipExtensionError):       # segment count must match ceil(Nk / seg_len)
        C.flash_attn_d64_static_partial(qkv, kk, vv, B, H, Nq, Nk, *strides, loose, o_p, l_p, c_p, 0, Wn + 1, seg_len=seg, skip_seg=1)


def test_flash_attn_static_key_split_fallback(C):
    """Rows below the acceptance threshold are flagged by the combine kernel and redone over all keys."""
    H, Nq, Nk = 2, 500, 4000
    Cdim = H * 64
    qkv = _rand((Nk, 3 * Cdim), 181, 1.0, F16)
    x = qkv.view(Nk, 3, H, 64)
    x[:, 0] *= 0.125 * 1.4426950408889634 * 3.0
    x[::3, 0] *= 0.01
    qkmax = torch.zeros(32, device="cuda")
    qkmax[:H] = x[:Nq, 0].float().norm(dim=-1).amax(0)
    qkmax[16:16 + H] = x[:, 1].float().norm(dim=-1).amax(0)
    flags = torch.zeros(H * 4, dtype=torch.int32, device="cuda")
    ws = torch.empty(C.static_attn_ws_bytes(1, H, Nq, Nk), dtype=torch.uint8, device="cuda")
    assert ws.numel() > 0
    o = torch.full((Nq, Cdim), float("nan"), dtype=F16, device="cuda")
    C.flash_attn_d64_static(qkv, qkv[:, Cdim:], qkv[:, 2 * Cdim:], o, 1, H, Nq, Nk, 0, 3 * Cdim, 0, 3 * Cdim, 0, 3 * Cdim,
                            0, Cdim, qkmax, flags, 0, ws)
    q, k, v = x[:Nq, 0].transpose(0, 1), x[:, 1].transpose(0, 1), x[:, 2].transpose(0, 1)
    ref = _attn_ref(q, k, v, 0.6931471805599453).transpose(0, 1).reshape(Nq, Cdim)
    mx, l2 = _relerr(o, ref)
    assert int(flags.sum()) > 0 and not torch.isnan(o.float()).any() and mx < 2e-3 and l2 < 5e-4, (int(flags.sum()), mx, l2)


def test_flash_attn_static_per_row_bound(C):
    """A few query rows with 30x the norm of the rest (register / camera tokens of a trained checkpoint): the shift is taken
    per query row (|q_i| max|k|, from the operand fragments), so the ordinary rows keep their tight bound -- only the tiles
    that CONTAIN an outlier row go to the online-max pass (a head-wide max|q| would send every tile there)."""
    H, N = 4, 6000
    Cdim = H * 64
    qkv = _rand((N, 3 * Cdim), 211, 1.0, F16)
    x = qkv.view(N, 3, H, 64)
    qk = x[:, :2].float()
    x[:, :2] = (qk / qk.norm(dim=-1, keepdim=True) * 8.0).half()
    x[:, 0] *= 0.125 * 1.4426950408889634 * 1.3
    outliers = [0, 1, 2, 3, 4, 1374, 1375, 2748, 5999]
    x[outliers, 0] *= 30.0
    tiles = {r // 256 for r in outliers}
    o, ref, flags = _static_attn(C, qkv, 1, H, N, N, N, 6256, F16)
    fl = flags[:-3].view(H, -1)
    nflag = int(fl.sum())
    report("attn_static_f16_per_row_bound", dict(flagged=nflag, tiles=int(fl.numel()), tiles_with_outliers=len(tiles) * H))
    assert 0 < nflag <= len(tiles) * H
    assert all(int(fl[:, t].sum()) == 0 for t in range(fl.shape[1]) if t not in tiles)
    mx, l2 = _relerr(o, ref)
    assert not torch.isnan(o.float()).any() and mx < 2e-3 and l2 < 5e-4, (mx, l2)
    for r in outliers + [5, 33, 1376]:
        assert _relerr(o[r], ref[r])[1] < 1e-3, r


def test_flash_attn_static_guard(C):
    """Adaptive switch: when more than 1/8 of the query tiles fail the acceptance test the call site skips the static kernel
    for the next 16 calls (online-max kernel only), then tries it again; a call site that was never measured inherits the
    verdict of the previous layer.  The result meets the tolerance in every state.  The losing input is an attention SINK:
    one key with 10x the norm of the rest sets max|k| -- the bound of every row -- while almost no row scores near it."""
    H, N = 2, 4000
    Cdim = H * 64
    good = _rand((N, 3 * Cdim), 233, 1.0, F16)
    xg = good.view(N, 3, H, 64)
    qk = xg[:, :2].float()
    xg[:, :2] = (qk / qk.norm(dim=-1, keepdim=True) * 8.0).half()
    xg[:, 0] *= 0.125 * 1.4426950408889634 * 1.3
    bad = good.clone()
    bad.view(N, 3, H, 64)[1234, 1] *= 10.0
    ntiles = H * ((N + 255) // 256)
    flags = torch.zeros(ntiles, dtype=torch.int32, device="cuda")

    def setup(data):
        x = data.view(N, 3, H, 64)
        qkmax = torch.zeros(32, device="cuda")
        qkmax[16:16 + H] = x[:, 1].float().norm(dim=-1).amax(0)
        q, k, v = x[:, 0].transpose(0, 1), x[:, 1].transpose(0, 1), x[:, 2].transpose(0, 1)
        return qkmax, _attn_ref(q, k, v, 0.6931471805599453).transpose(0, 1).reshape(N, Cdim)

    def run(data, qkmax, g, gp=None):
        o = torch.full((N, Cdim), float("nan"), dtype=F16, device="cuda")
        C.flash_attn_d64_static(data, data[:, Cdim:], data[:, 2 * Cdim:], o, 1, H, N, N, 0, 3 * Cdim, 0, 3 * Cdim, 0, 3 * Cdim,
                                0, Cdim, qkmax, flags, 6256, None, g, gp)
        return o

    def ok(o, ref):
        mx, l2 = _relerr(o, ref)
        assert not torch.isnan(o.float()).any() and mx < 2e-3 and l2 < 5e-4, (mx, l2)

    qm_bad, ref_bad = setup(bad)
    qm_good, ref_good = setup(good)
    guard = C.new_attn_guard("cuda")
    ok(run(bad, qm_bad, guard), ref_bad)
    st = guard.tolist()
    assert st[1] > ntiles // 8 and st[2] == ntiles and st[0] == 16 and st[3] == 1, st      # measured: static loses here
    for i in range(16):                               # 16 calls on the online-max kernel alone
        o = run(bad, qm_bad, guard)
        st = guard.tolist()
        assert st[1] == -1 and st[0] == 15 - i and int(flags.sum()) == ntiles, (i, st)
        if i in (0, 15):
            ok(o, ref_bad)
    ok(run(bad, qm_bad, guard), ref_bad)              # the static kernel is tried again -- and loses again
    st = guard.tolist()
    assert st[1] > ntiles // 8 and st[0] == 16, st
    # inheritance: a fresh call site behind a losing one skips at once; behind a winning one it measures itself
    fresh = C.new_attn_guard("cuda")
    ok(run(bad, qm_bad, fresh, guard), ref_bad)
    assert fresh.tolist()[:2] == [15, -1]
    winner, fresh2 = C.new_attn_guard("cuda"), C.new_attn_guard("cuda")
    ok(run(good, qm_good, winner), ref_good)
    assert winner.tolist()[:3] == [0, 0, ntiles]
    ok(run(good, qm_good, fresh2, winner), ref_good)
    assert fresh2.tolist()[:3] == [0, 0, ntiles]
    # without a guard nothing changes: static pass, every tile redone
    ok(run(bad, qm_bad, None), ref_bad)


def test_k_rownorm_max(C):
    """Largest per-head row norm of a strided 16-bit key matrix -> qkmax[16..31]; entries 0..15 stay as they were."""
    for dtype, rows in ((F16, 5003), (torch.bfloat16, 2), (F16, 1)):
        kv = _rand((rows, 2048), 77 + rows, 1.0, dtype)
        kv[rows // 2, 3 * 64:4 * 64] *= 5.0
        qkmax = torch.full((C.QKMAX_NUMEL,), -2.0, device="cuda")
        C.k_rownorm_max(kv[:, :1024], qkmax)
        ref = kv[:, :1024].float().view(rows, 16, 64).norm(dim=-1).amax(0)
        assert torch.allclose(qkmax[16:32], ref, rtol=1e-6, atol=0), (qkmax[16:32], ref)
        assert torch.all(qkmax[:16] == -2.0)


def test_qknorm_rope_prescale_and_norm_maxima(C):
    """q_scale is folded into q before the 16-bit rounding; qkmax receives the per-head maxima of the norms of the written
    q and k vectors (the inputs of the static softmax bound)."""
    from iggt_official_amd.layers.rope import RotaryPositionEmbedding2D

    S, gh, gw, psi = 3, 5, 7, 5
    P = psi + gh * gw
    T = S * P
    qkv = _rand((T, 3072), 150, 1.5, F16)
    qw, qb, kw, kb = (_rand((64,), 51) * 0.3 + 1, _rand((64,), 52, 0.2), _rand((64,), 53) * 0.3 + 1,
                      _rand((64,), 54, 0.2))
    cos, sin = RotaryPositionEmbedding2D(100).tables(64, max(gh, gw), torch.device("cuda"))
    plain, scaled = qkv.clone(), qkv.clone()
    C.qknorm_rope(plain, plain, plain[:, 1024:], None, qw, qb, kw, kb, cos, sin, T, P, gw, psi, 1e-5)
    qkmax = torch.full((C.QKMAX_NUMEL,), -1.0, device="cuda")
    sc = 0.125 * C.LOG2E
    C.qknorm_rope(scaled, scaled, scaled[:, 1024:], None, qw, qb, kw, kb, cos, sin, T, P, gw, psi, 1e-5, q_scale=sc,
                  qkmax=qkmax)
    assert torch.equal(scaled[:, 1024:], plain[:, 1024:])                      # k, v untouched by the scale
    mx, l2 = _relerr(scaled[:, :1024], plain[:, :1024].double() * sc)
    assert mx < 8e-4 and l2 < 4e-4, (mx, l2)                                   # one fp16 rounding apart
    qn = scaled[:, :1024].float().view(T, 16, 64).norm(dim=-1).amax(0)
    kn = scaled[:, 1024:2048].float().view(T, 16, 64).norm(dim=-1).amax(0)
    assert torch.allclose(qkmax[:16], qn, rtol=1e-6, atol=0) and torch.allclose(qkmax[16:32], kn, rtol=1e-6, atol=0)


def test_layernorm_f16_out(C):
    rows, Cdim = 1003, 1024
    x = _rand((rows, Cdim), 40, 3.0) + 0.7
    w, b = _rand((Cdim,), 41) * 0.1 + 1, _rand((Cdim,), 42, 0.1)
    big = torch.empty(rows, Cdim + 64, dtype=F16, device="cuda")
    out = big[:, :Cdim]     # padded row stride (layers/blocks.py ROW_PAD)
    C.layernorm(x, w, b, out, 1e-5)
    ref = torch.nn.functional.layer_norm(x.double(), (Cdim,), w.double(), b.double(), 1e-5)
    mx, l2 = _relerr(out, ref)
    assert mx < 7e-4 and l2 < 4e-4, (mx, l2)


def test_qknorm_rope_f16(C):
    from iggt_official_amd.layers.rope import RotaryPositionEmbedding2D

    S, gh, gw, psi = 2, 5, 7, 5
    P = psi + gh * gw
    T = S * P
    qkv = _rand((T, 3072), 50, 1.5, F16)
    orig = qkv.clone()
    qw, qb, kw, kb = (_rand((64,), 51) * 0.1 + 1, _rand((64,), 52, 0.1), _rand((64,), 53) * 0.1 + 1,
                      _rand((64,), 54, 0.1))
    cos, sin = RotaryPositionEmbedding2D(100).tables(64, max(gh, gw), torch.device("cuda"))
    vcopy = torch.empty(T, 1024, dtype=F16, device="cuda")
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
        assert mx < 8e-4 and l2 < 4e-4, (idx, mx, l2)  # one fp16 rounding of the result
    assert torch.equal(qkv[:, 2048:], orig[:, 2048:]) and torch.equal(vcopy, orig[:, 2048:])


def test_im2row_f16(C):
    S, H, W = 2, 28, 42
    img = torch.rand(S, 3, H, W, generator=torch.Generator().manual_seed(60)).cuda()
    gh, gw = H // 14, W // 14
    out = torch.empty(S * gh * gw, 640, dtype=F16, device="cuda")
    C.im2row_patch14(img, out, S, H, W, 640)
    mean = torch.tensor([0.485, 0.456, 0.406], device="cuda").view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device="cuda").view(1, 3, 1, 1)
    ref = torch.nn.functional.unfold((img - mean) / std, 14, stride=14).transpose(1, 2).reshape(S * gh * gw, 588)
    assert torch.all(out[:, 588:] == 0)
    assert (out[:, :588].float() - ref).abs().max() < 2e-3


@pytest.mark.parametrize("dtype", [F16, torch.bfloat16])
@pytest.mark.parametrize("rows,K,step", [(43, 1024, 1), (5000, 4096, 2), (1369, 640, 3), (7, 2048, 16)])
def test_colmean_and_bias_correct(C, dtype, rows, K, step):
    """Mean-input compensation kernels: mu = column mean over a row sample; out = bias + dW mu."""
    big = _rand((rows, K + 64), 90, 2.0, dtype) + 0.5
    x = big[:, :K]                                    # padded row stride
    mu = torch.full((K,), float("nan"), device="cuda")
    C.colmean(x, mu, step)
    ref_mu = x[::step].double().mean(0)
    assert _relerr(mu, ref_mu)[0] < 1e-5
    N = 384
    dw = _rand((N, K), 91, 1e-4, dtype)
    bias = _rand((N,), 92)
    out = torch.empty(N, device="cuda")
    C.bias_correct(dw, mu, bias, out)
    ref = bias.double() + dw.double() @ mu.double()
    assert float((out.double() - ref).abs().max()) < 1e-6
    C.bias_correct(dw, mu, None, out)
    assert float((out.double() - dw.double() @ mu.double()).abs().max()) < 1e-6


def test_qknorm_rope_head_group_layout(C):
    """k / v written in head-group layout [G][T][K(hg*64) | V(hg*64)] equal the flat outputs, regrouped."""
    from iggt_official_amd.layers.rope import RotaryPositionEmbedding2D

    S, gh, gw, psi = 2, 3, 4, 5
    P = psi + gh * gw
    T = S * P
    qkv = _rand((T, 3072), 55, 1.5, F16)
    qw, qb, kw, kb = (_rand((64,), 51) * 0.1 + 1, _rand((64,), 52, 0.1), _rand((64,), 53) * 0.1 + 1,
                      _rand((64,), 54, 0.1))
    cos, sin = RotaryPositionEmbedding2D(100).tables(64, max(gh, gw), torch.device("cuda"))
    flat = torch.empty(T, 2048, dtype=F16, device="cuda")
    q1 = qkv.clone()
    C.qknorm_rope(q1, q1, flat, flat[:, 1024:], qw, qb, kw, kb, cos, sin, T, P, gw, psi, 1e-5)
    for G in (2, 4, 16):
        hg = 16 // G
        D = 2 * hg * 64
        grp = torch.full((G, T, D), float("nan"), dtype=F16, device="cuda")
        q2 = qkv.clone()
        C.qknorm_rope(q2, q2, grp[0], grp[0][:, hg * 64:], qw, qb, kw, kb, cos, sin, T, P, gw, psi, 1e-5,
                      heads_per_group=hg, k_group_stride=T * D, v_group_stride=T * D)
        assert torch.equal(q2[:, :1024], q1[:, :1024])
        for g in range(G):
            assert torch.equal(grp[g][:, :hg * 64], flat[:, g * hg * 64:(g + 1) * hg * 64])
            assert torch.equal(grp[g][:, hg * 64:], flat[:, 1024 + g * hg * 64:1024 + (g + 1) * hg * 64])
