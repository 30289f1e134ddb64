#!/usr/bin/env python3
"""fc1 + GELU epilogue of the 256 x 256 GEMM: LDS-table GELU (round 4) against the polynomial erfc epilogue (IGGT_GELU_LUT=0),
same box, separate processes (the switch is read once per process).  Also the three other trunk shapes for reference.
Usage: python probes/gemm_gelu_ab.py [views ...] > profiles/r04_gemm_gelu_ab.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(views):
    sys.path.insert(0, ROOT)
    import torch

    from iggt_official_amd import _C

    _C.load()
    P, C = 1374, 1024
    for S in views:
        T = S * P
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
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    fn()
                e.record()
                e.synchronize()
                ts.append(s.elapsed_time(e) / 10)
            t = sorted(ts)[2] * 1e-3
            print(f"  views {S:3d} {name:5s} M={T:6d} N={N:5d} K={K:5d}: {t * 1e3:8.3f} ms  {2 * T * N * K / t / 1e12:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker([int(v) for v in sys.argv[2:]])
    else:
        views = sys.argv[1:] or ["32", "8"]
        for lut in ("1", "0"):
            print(f"# IGGT_GELU_LUT={lut} ({'LDS-table GELU' if lut == '1' else 'polynomial erfc GELU'}), fp16 operands", flush=True)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker"] + views, env=dict(os.environ, IGGT_GELU_LUT=lut))
