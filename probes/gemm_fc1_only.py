"""Run only the fc1 + GELU GEMM at 32 views, fp16 (for rocprofv3 --pmc passes; IGGT_GELU_LUT=0/1 selects the epilogue)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C
_C.load()
T, C = 32 * 1374, 1024
a = torch.randn(T, C, device="cuda").to(torch.float16)
w = (torch.randn(4 * C, C, device="cuda") * C ** -0.5).to(torch.float16)
b = torch.randn(4 * C, device="cuda")
out = torch.empty(T, 4 * C, dtype=torch.float16, device="cuda")
for _ in range(3):
    _C.gemm_h16(a, w, out, bias=b, act=1)
torch.cuda.synchronize()
print("done")
