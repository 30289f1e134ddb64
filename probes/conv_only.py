"""One hot conv shape of the DPT head in a loop (for rocprofv3 --pmc / timing): python probes/conv_only.py [N H Cin Cout K [prec]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iggt_official_amd import _C
from iggt_official_amd.heads import convops as co

N, H, Cin, Cout, K = (int(a) for a in (sys.argv[1:6] + ["8", "148", "256", "256", "3"][len(sys.argv[1:6]):]))
PREC = int(sys.argv[6]) if len(sys.argv) > 6 else 3
x = torch.randn(N, H, H, Cin, device="cuda")
conv = torch.nn.Conv2d(Cin, Cout, K, padding=K // 2).cuda()
pk = co.pack_conv2d(conv)
y = torch.empty(N, H, H, Cout, device="cuda")
fn = lambda: co.run(pk, x, out=y, prec=PREC)
for _ in range(3):
    fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    fn()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
fl = 2.0 * N * H * H * Cout * Cin * K * K
print(f"conv N={N} {H}x{H} {Cin}->{Cout} k{K} prec {PREC}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF/s useful ({PREC * fl / ms / 1e9:.0f} MFMA-equivalent)")
