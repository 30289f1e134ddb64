"""Every convolution shape of one DPT head at `views` x 518^2, timed at prec 3 (split-bf16, three MFMA passes) and prec 2
(fp16 hi + lo activations x fp16 weights + mean-input compensation, two passes):

    python probes/conv_prec_ab.py [views]            # wall time per call (HIP events, 10 calls)
    rocprofv3 --kernel-trace --stats -- python probes/conv_prec_ab.py   # per-kernel split (conv / chanmean / corr)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from iggt_official_amd.heads import convops as co

views = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = 37
# name, Cin, Cout, k, stride, map, count per head, relu_in
SHAPES = [("output_conv1", 256, 128, 3, 1, 8 * g, 1, False),
          ("refinenet1 rcu", 256, 256, 3, 1, 4 * g, 4, True), ("layer1_rn", 256, 256, 3, 1, 4 * g, 1, False),
          ("refinenet1.out_conv", 256, 256, 1, 1, 8 * g, 1, False),
          ("refinenet2 rcu", 256, 256, 3, 1, 2 * g, 4, True), ("layer2_rn", 512, 256, 3, 1, 2 * g, 1, False),
          ("refinenet2.out_conv", 256, 256, 1, 1, 4 * g, 1, False),
          ("refinenet3 rcu", 256, 256, 3, 1, g, 4, True), ("layer3_rn", 1024, 256, 3, 1, g, 1, False),
          ("refinenet3.out_conv", 256, 256, 1, 1, 2 * g, 1, False),
          ("refinenet4 rcu", 256, 256, 3, 1, 19, 2, True), ("layer4_rn", 1024, 256, 3, 1, 19, 1, False),
          ("refinenet4.out_conv", 256, 256, 1, 1, g, 1, False),
          ("resize_layers.3", 1024, 1024, 3, 2, g, 1, False)]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


tot = {2: 0.0, 3: 0.0}
print(f"{'layer':22s} {'shape':28s} {'x':>2s} {'prec3 ms':>9s} {'TF/s':>6s} {'prec2 ms':>9s} {'TF/s':>6s} {'ratio':>6s}")
for name, cin, cout, k, s, hw, cnt, relu in SHAPES:
    conv = torch.nn.Conv2d(cin, cout, k, s, k // 2).cuda()
    x = torch.randn(views, hw, hw, cin, device="cuda")
    pc = co.pack_conv2d(conv)
    ho = (hw + 2 * (k // 2) - k) // s + 1
    y = torch.empty(views, ho, ho, cout, device="cuda")
    fl = 2.0 * views * ho * ho * cout * cin * k * k
    ms = {p: timed(lambda p=p: co.run(pc, x, out=y, relu_in=relu, prec=p)) for p in (3, 2)}
    for p in (2, 3):
        tot[p] += cnt * ms[p]
    print(f"{name:22s} {f'{cin}->{cout} k{k}s{s} @{hw}^2':28s} {cnt:2d} {ms[3]:9.3f} {fl / ms[3] / 1e9:6.0f} {ms[2]:9.3f} {fl / ms[2] / 1e9:6.0f} "
          f"{ms[2] / ms[3]:6.2f}")
    del x, y, pc, conv
print(f"one head, these layers: prec 3 {tot[3]:.2f} ms, prec 2 {tot[2]:.2f} ms ({tot[2] / tot[3]:.2f})")
