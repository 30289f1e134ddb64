#!/usr/bin/env python3
"""Per-rank trunk GEMM shapes (M = 5 496: 4 views @ 518^2; M = 10 992: 8 views) under the dispatcher's switches, fp16:
default, IGGT_GEMM_DUO192=0 (256-row tiles only), IGGT_GEMM_DUO=1 (two-workgroups-per-CU kernel wherever it applies: proj too).
Prints time, rate and the worst relative error against an fp64 product on a row sample."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(views):
    sys.path.insert(0, ROOT)
    import torch

    from iggt_official_amd import _C

    _C.load()
    T, C = views * 1374, 1024
    for name, N, K, kw in [("qkv", 3 * C, C, {}), ("proj", C, C, {}), ("fc1", 4 * C, C, dict(act=1)), ("fc2", C, 4 * C, {})]:
        a = torch.randn(T, K, device="cuda").to(torch.float16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.float16)
        b = torch.randn(N, device="cuda")
        g = torch.randn(N, device="cuda")
        rows = torch.arange(0, T, 53, device="cuda")
        base = a[rows].double() @ w.double().t() + b.double()
        if name in ("proj", "fc2"):
            out = torch.zeros(T, N, device="cuda")
            fn = lambda: _C.gemm_h16(a, w, out, bias=b, gamma=g, accumulate=True)  # noqa: E731
            fn()
            err = float((out[rows].double() - g.double() * base).abs().max() / base.abs().max())
        else:
            out = torch.empty(T, N, dtype=torch.float16, device="cuda")
            fn = lambda: _C.gemm_h16(a, w, out, bias=b, **kw)  # noqa: E731
            fn()
            ref = torch.nn.functional.gelu(base) if kw else base
            err = float((out[rows].double() - ref).abs().max() / ref.abs().max())
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                fn()
            e.record()
            e.synchronize()
            ts.append(s.elapsed_time(e) / 20)
        t = sorted(ts)[3] * 1e-3
        print(f"   {name:5s} M={T} N={N:5d} K={K:5d}: {t * 1e6:7.1f} us  {2 * T * N * K / t / 1e12:7.1f} TF/s  err {err:.1e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]))
    else:
        for views in ([int(v) for v in sys.argv[1:]] or [4, 8]):
            for env in ({}, {"IGGT_GEMM_DUO192": "0"}, {"IGGT_GEMM_DUO": "1"}, {"IGGT_GEMM_DUO": "1", "IGGT_GEMM_DUO192": "0"}):
                print(f"# {views} views, {env or 'default'}", flush=True)
                subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", str(views)], env=dict(os.environ, **env))
