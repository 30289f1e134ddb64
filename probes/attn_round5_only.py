#!/usr/bin/env python3
"""A few launches of ONE of this round's global-attention forms at the bench shape (N = 43 968, 16 heads, fp16), for rocprofv3
counter passes (probes/pmc_kernel.sh OUT FILTER probes/attn_round5_only.py MODE [launches]):
  x3     flash_attn_x3_kernel: fp16 hi + lo operand pairs, three MFMA passes per product (csrc/x3.hip)
  est    the adaptive static-bound launch on the "affine" score regime (trained-like q/k-norm scales): after two calls the switch
         sits in estimated-shift mode, so the counters see the FINAL estimated-shift instantiation + its pre-pass / second chance
  rank   `est` at the per-rank shape of an 8-GPU run (Nq = 5 496, Nk = 43 968): the one-pass launch a sharded call site issues"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from iggt_official_amd import _C  # noqa: E402

_C.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, C, P, S = 16, 1024, 1374, int(os.environ.get("VIEWS", "32"))
T = S * P
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn(T, 3, H, 64, generator=g, device="cuda")
qk = x[:, :2]
qk = (qk - qk.mean(-1, keepdim=True)) / qk.std(-1, keepdim=True, unbiased=False)
if mode != "x3":
    gam = torch.exp(torch.randn(2, H, 64, generator=g, device="cuda"))
    qk = qk * (gam / gam.pow(2).mean(-1, keepdim=True).sqrt())[None]
x[:, :2] = qk
x[:, 0] *= 0.125 * _C.LOG2E * 1.3
if mode == "x3":
    src = x.reshape(T, 3 * C)
    buf = torch.empty(T, 6 * C, dtype=torch.float16, device="cuda")
    hi = src.half()
    buf[:, :3 * C] = hi
    buf[:, 3 * C:] = (src - hi.float()).half()
    o = torch.empty(T, 3 * C, dtype=torch.float16, device="cuda")
    for _ in range(n):
        _C.flash_attn_x3(buf, buf[:, 3 * C:], buf[:, C:], buf[:, 4 * C:], buf[:, 2 * C:], buf[:, 5 * C:], o, C, 1, H, T, T,
                         0, 6 * C, 0, 6 * C, 0, 6 * C, 0, 3 * C)
else:
    qkv = x.reshape(T, 3 * C).half()
    Nq = T // 8 if mode == "rank" else T
    q = qkv[3 * Nq:4 * Nq] if mode == "rank" else qkv
    qkmax = torch.zeros(_C.QKMAX_NUMEL, device="cuda")
    _C.k_rownorm_max(qkv[:, C:2 * C], qkmax)
    o = torch.empty(Nq, C, dtype=torch.float16, device="cuda")
    flags = torch.zeros(H * ((Nq + 127) // 128), dtype=torch.int32, device="cuda")
    est_ws = torch.zeros(_C.static_attn_est_ws_bytes(1, H, Nq, T), dtype=torch.uint8, device="cuda")
    guard = _C.new_attn_guard("cuda")
    for _ in range(n + 2):
        _C.flash_attn_d64_static(q, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, Nq, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C, qkmax, flags, 0,
                                 None, guard, None, est_ws=est_ws, key_period=P, key_nspecial=5)
    torch.cuda.synchronize()
    print("guard", guard.tolist())
torch.cuda.synchronize()
