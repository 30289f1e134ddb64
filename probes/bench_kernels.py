"""Micro-benchmarks of the hot kernels at the BASELINE shapes (GPU box).  Prints TF/s per kernel."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    _C.load()
    S_list = [int(x) for x in (sys.argv[1:] or ["8", "32"])]
    P, C, H = 1374, 1024, 16
    for S in S_list:
        T = S * P
        xn = torch.randn(T, C, device="cuda").to(torch.bfloat16)
        for name, N, K, kw in [("qkv", 3 * C, C, {}), ("proj", C, C, {}), ("fc1", 4 * C, C, dict(act=1)),
                               ("fc2", C, 4 * C, {})]:
            a = torch.randn(T, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
            b = torch.randn(N, device="cuda")
            if name in ("proj", "fc2"):
                out = torch.zeros(T, N, device="cuda")
                fn = lambda: _C.gemm_bf16(a, w, out, bias=b, gamma=b, accumulate=True)  # noqa: E731
            else:
                out = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
                fn = lambda: _C.gemm_bf16(a, w, out, bias=b, **kw)  # noqa: E731
            t = timeit(fn)
            print(f"S={S} gemm {name:5s} M={T} N={N} K={K}: {t*1e3:8.3f} ms  {2*T*N*K/t/1e12:7.1f} TF/s", flush=True)
        qkv = torch.randn(T, 3 * C, device="cuda").to(torch.bfloat16)
        o = torch.empty(T, C, dtype=torch.bfloat16, device="cuda")
        for tile in (5128, 6256):
            fn = lambda: _C.flash_attn_d64(qkv, qkv[:, C:], qkv[:, 2 * C:], o, S, H, P, P, P * 3 * C, 3 * C,  # noqa: E731
                                           P * 3 * C, 3 * C, P * 3 * C, 3 * C, P * C, C, 0.125, tile)
            t = timeit(fn)
            print(f"S={S} frame  attn tile={tile}: {t*1e3:8.3f} ms  {4*S*P*P*C/t/1e12:7.1f} TF/s", flush=True)
            fn = lambda: _C.flash_attn_d64(qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, T, T, 0, 3 * C,  # noqa: E731
                                           0, 3 * C, 0, 3 * C, 0, C, 0.125, tile)
            t = timeit(fn, iters=3, warm=1)
            print(f"S={S} global attn tile={tile}: {t*1e3:8.3f} ms  {4*T*T*C/t/1e12:7.1f} TF/s", flush=True)
        x = torch.randn(T, C, device="cuda")
        w1 = torch.ones(C, device="cuda")
        t = timeit(lambda: _C.layernorm(x, w1, w1, xn, 1e-5))
        print(f"S={S} layernorm: {t*1e6:8.1f} us  {T*C*6/t/1e12:6.2f} TB/s", flush=True)
        # stock PyTorch-ROCm SDPA on the same problem ("zero work" baseline, not part of the product)
        q = torch.randn(1, H, T, 64, device="cuda", dtype=torch.bfloat16)
        try:
            t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, q, q), iters=3, warm=1)
            print(f"S={S} torch SDPA global: {t*1e3:8.3f} ms  {4*T*T*C/t/1e12:7.1f} TF/s", flush=True)
        except Exception as ex:  # noqa: BLE001
            print("torch SDPA failed:", ex)


if __name__ == "__main__":
    main()
