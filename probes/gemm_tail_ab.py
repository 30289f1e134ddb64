#!/usr/bin/env python3
"""Per-rank trunk GEMM shapes (M = 5 496: 4 views @ 518^2) with and without the ragged-tail split (IGGT_GEMM_SPLIT_TAIL), fp16."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    sys.path.insert(0, ROOT)
    import torch

    from iggt_official_amd import _C

    _C.load()
    T, C = 4 * 1374, 1024
    for name, N, K, kw in [("qkv", 3 * C, C, {}), ("proj", C, C, {}), ("fc1", 4 * C, C, dict(act=1)), ("fc2", C, 4 * C, {})]:
        a = torch.randn(T, K, device="cuda").to(torch.float16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.float16)
        b = torch.randn(N, device="cuda")
        if name in ("proj", "fc2"):
            out = torch.zeros(T, N, device="cuda")
            fn = lambda: _C.gemm_h16(a, w, out, bias=b, gamma=b, accumulate=True)  # noqa: E731
        else:
            out = torch.empty(T, N, dtype=torch.float16, device="cuda")
            fn = lambda: _C.gemm_h16(a, w, out, bias=b, **kw)  # noqa: E731
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
        print(f"   {name:5s} M={T} N={N:5d} K={K:5d}: {t * 1e6:7.1f} us  {2 * T * N * K / t / 1e12:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker()
    else:
        for v in ("1", "0"):
            print(f"# IGGT_GEMM_SPLIT_TAIL={v}", flush=True)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker"], env=dict(os.environ, IGGT_GEMM_SPLIT_TAIL=v))
