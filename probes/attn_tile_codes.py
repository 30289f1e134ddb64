"""Static-bound global attention at the 32-view shape for every tile code (rows per workgroup x keys per macro tile)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C
_C.load()
H, C, P = 16, 1024, 1374
for views in (32, 8):
    T = views * P
    g = torch.Generator(device="cpu").manual_seed(1)
    qs = torch.randn(T, 3 * C, generator=g).half().cuda()
    qs[:, :C] *= 0.125 * _C.LOG2E
    x = qs.view(T, 3, H, 64)
    qkmax = torch.zeros(_C.QKMAX_NUMEL, device="cuda")
    qkmax[:16] = x[:, 0].float().norm(dim=-1).amax(0); qkmax[16:32] = x[:, 1].float().norm(dim=-1).amax(0)
    flags = torch.zeros(H * ((T + 127) // 128), dtype=torch.int32, device="cuda")
    o = torch.empty(T, C, dtype=torch.float16, device="cuda")
    for rep in range(2):
        for code in (6256, 5256, 6128, 5128):
            f = lambda: _C.flash_attn_d64_static(qs, qs[:, C:], qs[:, 2 * C:], o, 1, H, T, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C, qkmax, flags, code)
            f(); torch.cuda.synchronize(); ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
            ms = sorted(ts)[2]
            print(f"views={views} code={code}: {ms:.3f} ms  {4.0 * T * T * C / ms / 1e9:.0f} TF/s", flush=True)
