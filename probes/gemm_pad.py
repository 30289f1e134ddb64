"""Does a power-of-two operand row stride throttle the LDS-DMA GEMM?  Same GEMM, operands viewed out of padded buffers.
usage: python probes/gemm_pad.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iggt_official_amd import _C

dev = "cuda:0"
M = 43968


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def padded(rows, cols, pad, dtype):
    buf = (torch.rand(rows, cols + pad, device=dev) - 0.5).to(dtype)
    return buf[:, :cols]


for name, N, K, f32 in (("qkv", 3072, 1024, False), ("proj", 1024, 1024, True), ("fc1", 4096, 1024, False),
                        ("fc2", 1024, 4096, True)):
    for pa, pw, po in ((0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (128, 128, 0), (32, 32, 0),
                       (192, 192, 0)):
        a = padded(M, K, pa, torch.bfloat16)
        w = padded(N, K, pw, torch.bfloat16)
        out = padded(M, N, po, torch.float32 if f32 else torch.bfloat16)
        bias = torch.zeros(N, device=dev)
        gamma = torch.ones(N, device=dev) if f32 else None
        ms = bench(lambda: _C.gemm_bf16(a, w, out, bias=bias, gamma=gamma, accumulate=f32))
        print(f"{name} N={N} K={K} pad(a,w,o)=({pa},{pw},{po}): {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TF/s", flush=True)
