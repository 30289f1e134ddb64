#!/usr/bin/env python3
"""A/B of the global-attention kernels at the bench shapes: online-max (round 1) vs static-bound, for the in-tree library
and every probes/lib_alt/*.so (each in its own process: IGGT_HIP_LIB).  Interleaved rounds, median of HIP-event times."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch

    from iggt_official_amd import _C

    _C.load()
    H, C, P = 16, 1024, 1374
    res = {}
    for dt, name in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        for views, nq_views in ((32, 32), (32, 4), (8, 8)):
            T, Nq = views * P, nq_views * P
            g = torch.Generator(device="cpu").manual_seed(1)
            qkv = torch.randn(T, 3 * C, generator=g).to(dt).cuda()
            qs = qkv.clone()
            qs[:, :C] *= 0.125 * _C.LOG2E
            x = qs.view(T, 3, H, 64)
            qkmax = torch.zeros(32, device="cuda")
            qkmax[:16] = x[:, 0].float().norm(dim=-1).amax(0)
            qkmax[16:] = x[:, 1].float().norm(dim=-1).amax(0)
            flags = torch.zeros(H * ((T + 127) // 128), dtype=torch.int32, device="cuda")
            o = torch.empty(T, C, dtype=dt, device="cuda")
            fns = {
                "online": lambda: _C.flash_attn_d64(qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, Nq, T, 0, 3 * C, 0, 3 * C, 0,
                                                    3 * C, 0, C, 0.125, 0),
                "static": lambda: _C.flash_attn_d64_static(qs, qs[:, C:], qs[:, 2 * C:], o, 1, H, Nq, T, 0, 3 * C, 0, 3 * C,
                                                           0, 3 * C, 0, C, qkmax, flags, 0),
            }
            times = {k: [] for k in fns}
            for k, f in fns.items():
                f()
            torch.cuda.synchronize()
            for _ in range(7):
                for k, f in fns.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    f()
                    e1.record()
                    e1.synchronize()
                    times[k].append(e0.elapsed_time(e1))
            fl = 4.0 * Nq * T * C
            for k, v in times.items():
                v.sort()
                ms = v[len(v) // 2]
                res[f"{name} Nq={Nq} Nk={T} {k}"] = dict(ms=round(ms, 3), tflops=round(fl / ms / 1e9, 1), min_ms=round(v[0], 3))
            res[f"{name} Nq={Nq} Nk={T} flagged"] = int(flags.sum())
    print(json.dumps(res))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker()
    libs = [("in-tree", None)] + [(os.path.basename(p)[:-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "probes", "lib_alt", "*.so")))]
    for name, path in libs:
        env = dict(os.environ)
        if path:
            env["IGGT_HIP_LIB"] = path
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(name, "FAILED", out.stderr[-1500:])
            continue
        res = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"== {name}")
        for k, v in res.items():
            print(f"   {k:44s} {v}")


if __name__ == "__main__":
    main()
