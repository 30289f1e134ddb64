"""LayerNorm kernel rates: the trunk shape (C = 1024, fp32 -> fp16, 32 views) hot (the x a GEMM just wrote: Infinity-Cache resident)
and cold, the DPT-head shape (C = 2048, patch rows of a [S, P, 2C] kept layer -> fp16), and a plain copy of the same bytes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C
_C.load()


def med(f, n=9, flush=None):
    ts = []
    for _ in range(n):
        if flush is not None:
            flush.add_(1.0)                      # 1 GB pass: evicts the Infinity Cache
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2] * 1e3


S, P, g2 = 32, 1374, 1369
flush = torch.zeros(256 * 1024 * 1024, device="cuda")
for C, rows_in, rs, ro in ((1024, 0, 0, 0), (2048, g2, P, 5)):
    x = torch.randn(S * P, C, device="cuda")
    w, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    rows = S * P if rows_in == 0 else S * g2
    out = torch.empty(rows, C, dtype=torch.float16, device="cuda")
    f = lambda: _C.layernorm(x, w, b, out, 1e-5, rows=rows, rows_in=rows_in, rows_stride=rs, row_off=ro)
    byt = rows * C * 6
    hot, cold = med(f), med(f, flush=flush)
    y = torch.empty_like(x)
    cp_hot, cp_cold = med(lambda: y.copy_(x)), med(lambda: y.copy_(x), flush=flush)
    print(f"C={C}: LayerNorm {rows} rows hot {hot:.0f} us ({byt / hot / 1e6:.2f} TB/s), cold {cold:.0f} us ({byt / cold / 1e6:.2f} TB/s); "
          f"fp32 copy of the input hot {cp_hot:.0f} us ({x.numel() * 8 / cp_hot / 1e6:.2f} TB/s), cold {cp_cold:.0f} us "
          f"({x.numel() * 8 / cp_cold / 1e6:.2f} TB/s)", flush=True)
