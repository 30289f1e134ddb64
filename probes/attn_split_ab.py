#!/usr/bin/env python3
"""Per-rank global attention of an N-GPU run (Nq = 32/N views, Nk = 32 views): one pass (128-row tiles) vs key ranges."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C  # noqa: E402

_C.load()
H, C, P = 16, 1024, 1374
T = 32 * P
g = torch.Generator(device="cpu").manual_seed(1)
qkv = torch.randn(T, 3 * C, generator=g).to(torch.float16).cuda()
qk = qkv[:, :2 * C].float().view(T, 32, 64)
qkv[:, :2 * C] = (qk / qk.norm(dim=-1, keepdim=True) * 8.0).view(T, 2 * C).half()
qkv[:, :C] *= 0.125 * _C.LOG2E
x = qkv.view(T, 3, H, 64)
qkmax = torch.zeros(32, device="cuda")
qkmax[:16] = x[:, 0].float().norm(dim=-1).amax(0)
qkmax[16:] = x[:, 1].float().norm(dim=-1).amax(0)
flags = torch.zeros(H * ((T + 127) // 128), dtype=torch.int32, device="cuda")
o = torch.empty(T, C, dtype=torch.float16, device="cuda")


def t(fn, n=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for nrank in (8, 4, 2, 1):
    Nq = T // nrank
    fl = 4.0 * Nq * T * C
    one = t(lambda: _C.flash_attn_d64_static(qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, Nq, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C,
                                             qkmax, flags, 0, None))
    line = f"N={nrank} Nq={Nq}: one pass {one:.3f} ms ({fl / one / 1e9:.0f} TF/s)"
    nws = _C.static_attn_ws_bytes(1, H, Nq, T)
    if nws:
        ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
        sp = t(lambda: _C.flash_attn_d64_static(qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, Nq, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C,
                                                qkmax, flags, 0, ws))
        line += f" | auto split ({_C.attn_kernel_label(1, H, Nq, T, 'f16', True, 0, True)}) {sp:.3f} ms ({fl / sp / 1e9:.0f} TF/s)"
    for ks in (2, 3, 4, 5, 6, 8):
        o_part = torch.empty(ks, 1, Nq, C, dtype=torch.float16, device="cuda")
        l_part = torch.empty(ks, 1, H, Nq, device="cuda")

        def run():
            _C.flash_attn_d64_static_partial(qkv, qkv[:, C:], qkv[:, 2 * C:], 1, H, Nq, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, qkmax,
                                             o_part, l_part, 0, ks)
            _C.flash_attn_d64_static_combine(o_part, l_part, ks, qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, Nq, T, 0, 3 * C, 0,
                                             3 * C, 0, 3 * C, 0, C, flags)
        ms = t(run)
        line += f" | ks={ks}: {ms:.3f}"
    print(line, flush=True)
