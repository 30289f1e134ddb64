"""Run only the trunk GEMM shapes at 32 views (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C
_C.load()
T, C = 32 * 1374, 1024
for name, N, K in [("qkv", 3 * C, C), ("fc2", C, 4 * C)]:
    a = torch.randn(T, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    if name == "fc2":
        out = torch.zeros(T, N, device="cuda")
        for _ in range(3):
            _C.gemm_bf16(a, w, out, bias=b, gamma=b, accumulate=True)
    else:
        out = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
        for _ in range(3):
            _C.gemm_bf16(a, w, out, bias=b)
torch.cuda.synchronize()
print("done")
