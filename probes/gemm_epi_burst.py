"""Is the GEMM epilogue limited by the simultaneous write burst of all CUs?  One round of tiles at increasing CU counts."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C
_C.load()

def t(fn, n=15):
    fn(); fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]

K = 1024
for mode in ("h16", "h16gelu", "rmw"):
    for tiles in (8, 32, 64, 128, 256, 512):
        N = 1024
        M = tiles // 4 * 256
        a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / 32).half(); b = torch.randn(N, device="cuda")
        if mode == "rmw":
            out = torch.zeros(M, N, device="cuda"); fn = lambda: _C.gemm_h16(a, w, out, bias=b, gamma=b, accumulate=True)
        else:
            out = torch.empty(M, N, dtype=torch.float16, device="cuda"); fn = lambda: _C.gemm_h16(a, w, out, bias=b, act=int(mode == "h16gelu"))
        print(f"{mode:8s} tiles={tiles:4d}: {t(fn) * 1e3:7.1f} us", flush=True)
