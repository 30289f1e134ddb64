"""GEMM micro-benchmark at the trunk shapes (fp16 operands, padded A rows like the model): M = views * 1374."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C  # noqa: E402

_C.load()
P, C = 1374, 1024


def t(fn, n=9):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for S in [int(x) for x in (sys.argv[1:] or ["32", "4"])]:
    T = S * P
    for name, N, K, kw in [("qkv", 3 * C, C, {}), ("proj", C, C, {}), ("fc1", 4 * C, C, dict(act=1)), ("fc2", C, 4 * C, {})]:
        a = torch.randn(T, K + 64, device="cuda").half()[:, :K]
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        b = torch.randn(N, device="cuda")
        if name in ("proj", "fc2"):
            out = torch.zeros(T, N, device="cuda")
            fn = lambda: _C.gemm_h16(a, w, out, bias=b, gamma=b, accumulate=True)  # noqa: E731
        else:
            out = torch.empty(T, N, dtype=torch.float16, device="cuda")
            fn = lambda: _C.gemm_h16(a, w, out, bias=b, **kw)  # noqa: E731
        ms = t(fn)
        print(f"S={S} gemm {name:5s} M={T} N={N} K={K}: {ms:8.3f} ms  {2 * T * N * K / ms / 1e9:7.1f} TF/s", flush=True)

# cross-check (NOT part of the product): what the vendor library reaches on the same shapes (fp16 in / fp16 out, no epilogue)
if os.environ.get("IGGT_GEMM_XCHECK", "0") == "1":
    for S in (32, 4):
        T = S * P
        for name, N, K in [("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)]:
            a = torch.randn(T, K, device="cuda").half()
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
            ms = t(lambda: torch.mm(a, w.t()))
            print(f"S={S} hipBLASLt {name:5s} M={T} N={N} K={K}: {ms:8.3f} ms  {2 * T * N * K / ms / 1e9:7.1f} TF/s", flush=True)
