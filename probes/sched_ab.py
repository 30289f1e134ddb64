"""One line per hot kernel (median of 7 launches, HIP events): global attention static fp16 / bf16 at the 32-view shape, the four trunk
GEMM shapes with their production epilogues, one two-pass head convolution.  Run once per library build:
    for v in "" maxilp iterilp; do IGGT_HIP_LIB=${v:+probes/lib_alt/sched_$v.so} python probes/sched_ab.py; done
(probes/build_alt.py sched_*: the same sources under different LLVM scheduling strategies)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C
from iggt_official_amd.heads import convops as co
_C.load()
tag = os.path.basename(os.environ.get("IGGT_HIP_LIB", "production"))


def med(f, n=7):
    f(); f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2]


S, P, C, H = 32, 1374, 1024, 16
T = S * P
out = []
for dt, nm in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
    g = torch.Generator(device="cpu").manual_seed(1)
    qkv = torch.randn(T, 3 * C, generator=g).to(dt).cuda()
    o = torch.empty(T, C, dtype=dt, device="cuda")
    qkv[:, :C] *= 0.125 * _C.LOG2E
    x = qkv.view(T, 3, H, 64)
    qkmax = torch.zeros(_C.QKMAX_NUMEL, device="cuda")
    qkmax[:16] = x[:, 0].float().norm(dim=-1).amax(0); qkmax[16:32] = x[:, 1].float().norm(dim=-1).amax(0)
    flags = torch.zeros(H * ((T + 127) // 128), dtype=torch.int32, device="cuda")
    ms = med(lambda: _C.flash_attn_d64_static(qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, T, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C, qkmax, flags, 0))
    out.append(f"attn_{nm} {ms:.3f} ms {4.0 * T * T * C / ms / 1e9:.0f} TF/s")
    del qkv, o
dt = torch.float16
for name, N, K in [("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)]:
    pad = 64 if name == "qkv" else 0
    a = torch.zeros(T, K + pad, device="cuda", dtype=dt)[:, :K]; a.copy_(torch.randn(T, K, device="cuda"))
    w = torch.zeros(N, K + pad, device="cuda", dtype=dt)[:, :K]; w.copy_(torch.randn(N, K, device="cuda") * K ** -0.5)
    b = torch.randn(N, device="cuda")
    if name in ("proj", "fc2"):
        o2 = torch.zeros(T, N, device="cuda")
        ms = med(lambda: _C.gemm_h16(a, w, o2, bias=b, gamma=b, accumulate=True))
    else:
        o2 = torch.empty(T, N, dtype=dt, device="cuda")
        ms = med(lambda: _C.gemm_h16(a, w, o2, bias=b, act=1 if name == "fc1" else 0))
    out.append(f"{name} {ms*1e3:.0f} us {2.0 * T * N * K / ms / 1e9:.0f} TF/s")
    del a, w, o2
xc = torch.randn(32, 148, 148, 256, device="cuda")
conv = torch.nn.Conv2d(256, 256, 3, padding=1).cuda()
pk = co.pack_conv2d(conv)
y = torch.empty(32, 148, 148, 256, device="cuda")
ms = med(lambda: co.run(pk, xc, out=y, prec=2))
out.append(f"conv148_p2 {ms:.3f} ms {2.0 * 32 * 148 * 148 * 256 * 256 * 9 / ms / 1e9:.0f} TF/s")
print(f"{tag:24s} | " + " | ".join(out), flush=True)
