"""A/B of the global-attention kernel at the 32-view shape: python probes/attn_ab.py [f16|bf16] (env IGGT_ATTN_OPT=0/1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iggt_official_amd import _C

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
S, P, C, H = 32, 1374, 1024, 16
T = S * P
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(T, 3 * C, device="cuda", generator=g)
# realistic score spread: unit-variance q/k after the q/k-LayerNorm (scores ~ N(0, 64 * 0.125^2 = 1))
qkv = x.to(dt)
o = torch.empty(T, C, dtype=dt, device="cuda")
fn = lambda: _C.flash_attn_d64(qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, T, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C, 0.125, 0)
for _ in range(2):
    fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    fn()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print(f"{dt} OPT={os.environ.get('IGGT_ATTN_OPT', '1')}: {ms:.3f} ms  {4.0 * T * T * C / ms / 1e9:.1f} TF/s", flush=True)
