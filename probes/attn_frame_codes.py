"""Static-bound FRAME attention (32 x 16 sequences of 1 374 tokens) and the DINOv2 shape (1 370 tokens, online-max kernel) for every
tile code (rows per workgroup x keys per macro tile): which tiling the dispatcher should pick for short sequences."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C
_C.load()
H, C = 16, 1024
for views, P in ((32, 1374), (4, 1374), (32, 1370)):
    T = views * P
    g = torch.Generator(device="cpu").manual_seed(1)
    qs = torch.randn(T, 3 * C, generator=g).half().cuda()
    qs[:, :C] *= 0.125 * _C.LOG2E
    x = qs.view(T, 3, H, 64)
    qkmax = torch.zeros(_C.QKMAX_NUMEL, device="cuda")
    qkmax[:16] = x[:, 0].float().norm(dim=-1).amax(0); qkmax[16:32] = x[:, 1].float().norm(dim=-1).amax(0)
    flags = torch.zeros(views * H * ((P + 127) // 128), dtype=torch.int32, device="cuda")
    o = torch.empty(T, C, dtype=torch.float16, device="cuda")
    for code in (5128, 6128, 5256, 6256):
        f = lambda: _C.flash_attn_d64_static(qs, qs[:, C:], qs[:, 2 * C:], o, views, H, P, P, P * 3 * C, 3 * C, P * 3 * C, 3 * C,
                                             P * 3 * C, 3 * C, P * C, C, qkmax, flags, code)
        f(); torch.cuda.synchronize(); ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[3]
        print(f"static  views={views} P={P} code={code}: {ms*1e3:.0f} us  {4.0 * views * P * P * C / ms / 1e9:.0f} TF/s", flush=True)
    for code in (5128, 6128, 5256, 6256):   # online-max kernel (the DINOv2 blocks have no q / k norm, hence no static bound)
        f = lambda: _C.flash_attn_d64(qs, qs[:, C:], qs[:, 2 * C:], o, views, H, P, P, P * 3 * C, 3 * C, P * 3 * C, 3 * C,
                                      P * 3 * C, 3 * C, P * C, C, 1.0 / _C.LOG2E, code)
        f(); torch.cuda.synchronize(); ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[3]
        print(f"dynamic views={views} P={P} code={code}: {ms*1e3:.0f} us  {4.0 * views * P * P * C / ms / 1e9:.0f} TF/s", flush=True)
