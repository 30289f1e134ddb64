"""fp32 Linear (csrc/smallops.hip) at the camera-head shapes: S = 32 tokens against 2048-wide weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iggt_official_amd import _C
_C.load()
def t(fn, n=7, burst=40):
    """GPU time per call: bursts of back-to-back launches behind a long dummy kernel, so the queue never runs dry
    (a single call is shorter than its own host-side launch overhead)."""
    fn(); fn(); torch.cuda.synchronize(); ts = []
    big = torch.empty(1 << 28, device="cuda")
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        big.zero_()                      # ~1 ms of GPU work: the launches below queue up behind it
        e0.record()
        for _ in range(burst):
            fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / burst)
    return sorted(ts)[len(ts) // 2]
tot = 0
for name, N, K in (("qkv", 6144, 2048), ("proj", 2048, 2048), ("fc1", 8192, 2048), ("fc2", 2048, 8192), ("adaLN", 6144, 2048), ("pose", 1024, 2048)):
    # rotate through several weight copies so that the 256 MB of L2 + MALL do not hold them between calls
    ws = [torch.randn(N, K, device="cuda") for _ in range(max(2, int(600e6 / (N * K * 4))))]
    x = torch.randn(32, K, device="cuda"); b = torch.randn(N, device="cuda"); i = [0]; out = torch.empty(32, N, device="cuda")
    def f():
        i[0] = (i[0] + 1) % len(ws)
        _C.linear_f32(x, ws[i[0]], b, out=out)
    ms = t(f); tot += ms
    print(f"{name:6s} 32 x {N} x {K}: {ms * 1e3:7.1f} us   {N * K * 4 / ms / 1e9:6.2f} TB/s", flush=True)
print(f"sum {tot * 1e3:.1f} us")
