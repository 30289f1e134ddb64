import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C
_C.load()
P, C, H = 1374, 1024, 16
T = 32 * P
qkv = torch.randn(T, 3 * C, device="cuda").to(torch.bfloat16)
o = torch.empty(T, C, dtype=torch.bfloat16, device="cuda")
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
for views in (4, 8, 16, 32):
    nq = views * P
    for tile in (5128, 6256, 0):
        ms = t(lambda: _C.flash_attn_d64(qkv, qkv[:, C:], qkv[:, 2*C:], o, 1, H, nq, T, 0, 3*C, 0, 3*C, 0, 3*C, 0, C, 0.125, tile))
        print(f"global Nq={nq} ({views} local views) Nk={T} tile={tile}: {ms:.3f} ms {4*nq*T*C/ms/1e9:.0f} TF/s")
for views in (4, 8, 32):
    for tile in (5128, 6256, 0):
        ms = t(lambda: _C.flash_attn_d64(qkv, qkv[:, C:], qkv[:, 2*C:], o, views, H, P, P, P*3*C, 3*C, P*3*C, 3*C, P*3*C, 3*C, P*C, C, 0.125, tile))
        print(f"frame B={views} tile={tile}: {ms:.3f} ms {4*views*P*P*C/ms/1e9:.0f} TF/s")
