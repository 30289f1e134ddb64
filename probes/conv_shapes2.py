"""3x3 convolution micro-benchmark at the DPT head shapes (32 views): halo kernel vs GEMM-shaped kernel
(IGGT_CONV_HALO=0 in a second process)."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker():
    import torch.nn as nn

    from iggt_official_amd.heads import convops as co

    def t(fn, n=5):
        fn(); fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    for views, hw, cin, cout in ((32, 148, 256, 256), (32, 296, 256, 128), (32, 74, 256, 256), (32, 74, 512, 256), (8, 144, 256, 256),
                                 (8, 288, 256, 128), (8, 72, 256, 256), (4, 148, 256, 256)):
        conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda()
        pc = co.pack_conv2d(conv)
        x = torch.randn(views, hw, hw, cin, device="cuda")
        y = torch.empty(views, hw, hw, cout, device="cuda")
        ms = t(lambda: co.run(pc, x, out=y, relu_in=True))
        fl = 2.0 * views * hw * hw * cin * 9 * cout
        print(f"{views:3d} x {hw:3d}^2 {cin:4d}->{cout:3d}: {ms:7.3f} ms  {fl / ms / 1e9:6.0f} TF/s useful  {3 * fl / ms / 1e9:6.0f} MFMA-equivalent",
              flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker()
    else:
        for halo, tile in (("1", ""), ("1", "8x32"), ("1", "16x16"), ("0", "")):
            print(f"== IGGT_CONV_HALO={halo} IGGT_CONV_HALO_TILE={tile or 'auto'}", flush=True)
            env = dict(os.environ, IGGT_CONV_HALO=halo)
            if tile:
                env["IGGT_CONV_HALO_TILE"] = tile
            subprocess.run([sys.executable, os.path.abspath(__file__), "w"], env=env)
