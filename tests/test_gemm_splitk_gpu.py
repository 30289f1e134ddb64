"""Two-slice split-K of the two-workgroups-per-CU GEMM (GPU; round 5; csrc/gemm_bf16_duo.hip, include/iggt_hip.h iggt_gemm_*_ws).

At the per-rank shapes of an 8-GPU run (M = 5 496) every output tile of qkv / proj / fc2 is computed by two workgroups over half
of K each; the second to finish adds the first one's accumulators (fp32 slab, agent-scope release / acquire hand-over) and runs
the epilogue.  Checked: against fp64 at the tolerance of the one-workgroup kernel, bitwise equality with the un-split result's
epilogue semantics across repeated launches (which workgroup reduces varies; a + b does not), every epilogue mode the block
engine uses (16-bit out with bias / GELU; fp32 residual accumulate with LayerScale), ragged M, the workspace left clean."""
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


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,M,N,K,kind", [("qkv", 5496, 3072, 1024, "h16"), ("proj", 5496, 1024, 1024, "acc"),
                                             ("fc2", 5496, 1024, 4096, "acc"), ("fc1", 5496, 4096, 1024, "gelu"),
                                             ("ragged", 1379, 1024, 2048, "acc"), ("two views", 2748, 3072, 1024, "h16")])
def test_splitk_gemm_matches_fp64_and_is_deterministic(C, dt, name, M, N, K, kind):
    a = _rand((M, K), 1).to(dt)
    w = _rand((N, K), 2, K ** -0.5).to(dt)
    bias, gamma = _rand((N,), 3), _rand((N,), 4) * 0.1 + 1
    ws = torch.zeros(C.gemm_ws_bytes(M, N), dtype=torch.uint8, device="cuda")
    x0 = _rand((M, N), 5)
    prod = a.double() @ w.double().t() + bias.double()

    def run(ws_):
        if kind == "acc":
            out = x0.clone()
            C.gemm_h16(a, w, out, bias=bias, gamma=gamma, accumulate=True, ws=ws_)
        else:
            out = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
            C.gemm_h16(a, w, out, bias=bias, act=1 if kind == "gelu" else 0, ws=ws_)
        return out

    ref = x0.double() + gamma.double() * prod if kind == "acc" else (torch.nn.functional.gelu(prod) if kind == "gelu" else prod)
    plain = run(None)
    outs = [run(ws) for _ in range(6)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])                       # bitwise, whichever workgroup reduced
    e_split = float((outs[0].double() - ref).norm() / ref.norm())
    e_plain = float((plain.double() - ref).norm() / ref.norm())
    tol = 2e-6 if kind == "acc" else (4e-4 if dt == torch.float16 else 3e-3)
    assert e_split < tol and e_split < 1.5 * e_plain + 1e-7, (e_split, e_plain)
    head = ws[:64 + 4096 * 8].view(torch.int32)
    assert int(head.abs().sum()) == 0                        # ticket / ready words reset, error word never set
    report(f"gemm_splitk/{name}_{'f16' if dt == torch.float16 else 'bf16'}", dict(l2_split=e_split, l2_plain=e_plain))


def test_splitk_gemm_timing_report(C):
    """HIP-event times of the per-rank GEMMs with and without the workspace (reported; the emulated-rank bench is the gate that
    matters)."""
    M = 5496
    res = {}
    for name, N, K, kind in (("qkv", 3072, 1024, "h16"), ("proj", 1024, 1024, "acc"), ("fc1", 4096, 1024, "gelu"),
                             ("fc2", 1024, 4096, "acc")):
        a = _rand((M, K), 1).half()
        w = _rand((N, K), 2, K ** -0.5).half()
        bias, gamma = _rand((N,), 3), _rand((N,), 4) * 0.1 + 1
        ws = torch.zeros(C.gemm_ws_bytes(M, N), dtype=torch.uint8, device="cuda")
        out = torch.zeros(M, N, device="cuda") if kind == "acc" else torch.empty(M, N, dtype=torch.float16, device="cuda")
        for label, ws_ in (("plain", None), ("split", ws)):
            kw = dict(bias=bias, gamma=gamma, accumulate=True) if kind == "acc" else dict(bias=bias, act=1 if kind == "gelu" else 0)
            for _ in range(5):
                C.gemm_h16(a, w, out, ws=ws_, **kw)
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    C.gemm_h16(a, w, out, ws=ws_, **kw)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) / 20)
            t = sorted(ts)[3]
            res[f"{name}_{label}"] = dict(us=t * 1e3, tflops=2.0 * M * N * K / (t * 1e-3) / 1e12)
    report("gemm_splitk/timing_M5496", res)
    print(res)
