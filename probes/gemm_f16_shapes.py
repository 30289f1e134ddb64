"""The four trunk GEMM shapes in fp16 (production operand format) with their production epilogues, for rocprofv3 passes.
Usage: python probes/gemm_f16_shapes.py [views=32] [which=qkv,proj,fc1,fc2] [reps=3]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C
_C.load()
views = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = (sys.argv[2] if len(sys.argv) > 2 else "qkv,proj,fc1,fc2").split(",")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
T, C = views * 1374, 1024
dt = torch.float16
for name, N, K in [("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)]:
    if name not in which:
        continue
    pad = 64 if name == "qkv" else 0                     # layers/blocks.py ROW_PAD on the qkv operands
    a = torch.zeros(T, K + pad, device="cuda", dtype=dt)[:, :K]
    a.copy_(torch.randn(T, K, device="cuda"))
    w = torch.zeros(N, K + pad, device="cuda", dtype=dt)[:, :K]
    w.copy_(torch.randn(N, K, device="cuda") * K ** -0.5)
    b = torch.randn(N, device="cuda")
    if name in ("proj", "fc2"):
        out = torch.zeros(T, N, device="cuda")
        for _ in range(reps):
            _C.gemm_h16(a, w, out, bias=b, gamma=b, accumulate=True)
    else:
        out = torch.empty(T, N, dtype=dt, device="cuda")
        for _ in range(reps):
            _C.gemm_h16(a, w, out, bias=b, act=1 if name == "fc1" else 0)
torch.cuda.synchronize()
print("done")
