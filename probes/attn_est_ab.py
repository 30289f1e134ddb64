#!/usr/bin/env python3
"""A/B of the estimated-shift instantiation of the static attention kernel (probes/build_alt.py est_* variants vs the in-tree
library): est forced, fp16, N = 43 968, the four score regimes of probes/attn_static_robustness.py; median of 7 launches and the largest
difference to the online-max kernel's output.  Usage: python probes/attn_est_ab.py > profiles/r04_attn_est_ab.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "probes"))
    import torch

    import attn_static_robustness as r
    from iggt_official_amd import _C

    H, C, P, T = r.H, r.C, r.P, r.T
    for kind in ("noise", "affine", "sinks", "registers"):
        qkv, qkmax = r.make(kind, torch.float16)
        o = torch.empty(T, C, dtype=torch.float16, device="cuda")
        o_ref = torch.empty_like(o)
        flags = torch.zeros(H * ((T + 127) // 128), dtype=torch.int32, device="cuda")
        args = (qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, T, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C)
        _C.flash_attn_d64(qkv, qkv[:, C:], qkv[:, 2 * C:], o_ref, 1, H, T, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C, 0.6931471805599453, 0)
        est_ws = torch.zeros(_C.static_attn_est_ws_bytes(1, H, T, T), dtype=torch.uint8, device="cuda")

        def est():
            _C.flash_attn_d64_static(*args, qkmax, flags, 0, None, None, None, est_ws=est_ws, key_period=P, key_nspecial=5, est_mode=1)

        def classic():
            _C.flash_attn_d64_static(*args, qkmax, flags, 0, None, None, None)

        est()
        torch.cuda.synchronize()
        diff = float((o.float() - o_ref.float()).abs().max())
        rows = int(_C.static_attn_est_views(est_ws, 1, H, T)["rowcount"].sum())
        t_est = sorted(r.timed(est, 7))[3]
        t_cl = sorted(r.timed(classic, 7))[3] if kind == "noise" else float("nan")
        print(f"   {kind:8s} est forced {t_est:7.3f} ms   norm bound {t_cl:7.3f} ms   rows handed over {rows:6d}   max |o - o_online| {diff:.2e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker()
    else:
        libs = [("in-tree (max-ilp, per-block delta)", None)]
        alt = os.path.join(ROOT, "probes", "lib_alt")
        for n in sorted(os.listdir(alt)) if os.path.isdir(alt) else []:
            if n.startswith("est_") and n.endswith(".so"):
                libs.append((n[:-3], os.path.join(alt, n)))
        for name, path in libs:
            print(f"# {name}", flush=True)
            env = dict(os.environ)
            if path:
                env["IGGT_HIP_LIB"] = path
            subprocess.call([sys.executable, os.path.abspath(__file__), "--worker"], env=env)
