#!/usr/bin/env python3
"""Static-bound softmax on score statistics that are NOT LayerNorm-of-noise: flagged-tile fraction and wall
time of the global attention at the bench shape (N = 43 968, 16 heads, fp16 and bf16) for

  noise      q, k = per-head LayerNorm of Gaussian noise (|q^| = 1.875, |k| = 8): what every hash-noise fixture produces
  affine     trained-like q/k-norm affines: per-channel gamma log-normal (sigma = 1), drawn per head, on q and on k
  sinks      affine + 8 "attention sink" keys with 10x the norm of the rest (they set max|k|, the bound of every row)
  registers  affine + the 5 special-token rows of every view with 30x the query norm of the patch rows

in these launch modes: online-max kernel alone (the fallback everything is measured against); static norm bound without the
adaptive switch (static pass, then every flagged tile again); round 3's adaptive launch (norm bound or online-max only: no
workspace), steady state = mean over 34 calls (two retry periods); round 4: the ESTIMATED shift forced (pre-pass + static
kernel + row-granular redo, csrc/attention_est.hip) and the round-4 adaptive launch (norm bound -> estimated shift -> online-max
only) from a cold call site: first call, second call, steady state, with the mode the switch settled in and the rows / tiles
handed to the online-max pass.
Usage: python probes/attn_static_robustness.py > profiles/r04_attn_static_robustness.txt   (on the GPU box)
       python probes/attn_static_robustness.py --rank [W]   (round 5) the ONE-PASS launch a rank of a W-GPU run (default 8) issues once
       its call site is on the estimated shift: Nq = T / W query rows (rank W / 2 - 1's) against all T keys -- the same table."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from iggt_official_amd import _C  # noqa: E402

_C.load()
H, C, P, VIEWS = 16, 1024, 1374, int(os.environ.get("VIEWS", "32"))
T = VIEWS * P
WORLD = (int(sys.argv[sys.argv.index("--rank") + 1]) if len(sys.argv) > sys.argv.index("--rank") + 1 else 8) if "--rank" in sys.argv else 1
NQ, Q0 = T // WORLD, (T // WORLD) * max(WORLD // 2 - 1, 0)     # this rank's query rows: [Q0, Q0 + NQ)


def make(kind, dt, seed=3):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(T, 3, H, 64, generator=g, device="cuda")
    qk = x[:, :2]
    qk = (qk - qk.mean(-1, keepdim=True)) / qk.std(-1, keepdim=True, unbiased=False)       # per-head LayerNorm: |.| = 8
    if kind != "noise":
        gam = torch.exp(torch.randn(2, H, 64, generator=g, device="cuda"))                  # log-normal, sigma = 1
        gam = gam / gam.pow(2).mean(-1, keepdim=True).sqrt()                                 # rms 1 per head: typical |.| stays 8
        qk = qk * gam[None]
    x[:, :2] = qk
    x[:, 0] *= 0.125 * _C.LOG2E * 1.3
    if kind == "sinks":
        idx = torch.randperm(T, generator=g, device="cuda")[:8]
        x[idx, 1] *= 10.0
    if kind == "registers":
        rows = (torch.arange(VIEWS, device="cuda")[:, None] * P + torch.arange(5, device="cuda")[None]).reshape(-1)
        x[rows, 0] *= 30.0
    qkv = x.reshape(T, 3 * C).to(dt)
    qkmax = torch.zeros(_C.QKMAX_NUMEL, device="cuda")
    _C.k_rownorm_max(qkv[:, C:2 * C], qkmax)
    return qkv, qkmax


def timed(fn, reps):
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1))
    return out


def main():
    print(f"# global attention, {VIEWS} views @ 518^2: Nq = {NQ}" + (f" (rank {max(WORLD // 2 - 1, 0)} of {WORLD})" if WORLD > 1 else "")
          + f", Nk = {T}, 16 heads x 64; {torch.cuda.get_device_name(0)}")
    print(f"# {'operands':8s} {'input':10s} {'flagged tiles':>16s} {'online-max':>11s} {'norm bound,no sw':>17s} "
          f"{'r3 switch steady':>17s} {'est forced':>11s} {'r4: 1st / 2nd / steady':>26s} {'steady/online':>14s}  r4 state")
    for dt, name in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        for kind in ("noise", "affine", "sinks", "registers"):
            qkv, qkmax = make(kind, dt)
            o = torch.empty(NQ, C, dtype=dt, device="cuda")
            flags = torch.zeros(H * ((NQ + 127) // 128), dtype=torch.int32, device="cuda")
            args = (qkv[Q0:Q0 + NQ], qkv[:, C:], qkv[:, 2 * C:], o, 1, H, NQ, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C)

            def online():
                # q already carries scale * log2 e: softmax scale ln 2 in the online-max kernel's convention
                _C.flash_attn_d64(*args, 0.6931471805599453, 0)

            def static(guard=None):
                _C.flash_attn_d64_static(*args, qkmax, flags, 0, None, guard, None)

            online(), static()
            t_on = sorted(timed(online, 7))[3]
            t_st = sorted(timed(static, 7))[3]
            ntiles = H * ((NQ + 255) // 256)
            nflag = int(flags[:ntiles].sum())
            guard = _C.new_attn_guard("cuda")
            timed(lambda: static(guard), 1)
            steady = timed(lambda: static(guard), 34)
            t_r3 = sum(steady) / len(steady)
            est_ws = torch.zeros(_C.static_attn_est_ws_bytes(1, H, NQ, T), dtype=torch.uint8, device="cuda")

            def est(guard=None, force=1):
                _C.flash_attn_d64_static(*args, qkmax, flags, 0, None, guard, None, est_ws=est_ws, key_period=P, key_nspecial=5,
                                         est_mode=force)

            est()
            t_est = sorted(timed(est, 7))[3]
            views = _C.static_attn_est_views(est_ws, 1, H, NQ, T)
            rows_forced = int(views["rowcount"].sum())
            hi = views["hicount"].tolist()
            g4 = _C.new_attn_guard("cuda")
            t1 = timed(lambda: est(g4, 0), 1)[0]
            t2 = timed(lambda: est(g4, 0), 1)[0]
            steady = timed(lambda: est(g4, 0), 34)
            t_r4 = sum(steady) / len(steady)
            st = g4.tolist()
            mode = "online-max only" if st[0] > 0 or st[1] < 0 else ("estimated shift" if st[4] == 1 else "norm bound")
            print(f"  {name:8s} {kind:10s} {nflag:7d} / {ntiles:5d} {t_on:9.2f}ms {t_st:15.2f}ms {t_r3:15.2f}ms {t_est:9.2f}ms "
                  f"{t1:8.2f} /{t2:6.2f} /{t_r4:6.2f}ms {t_r4 / t_on:12.3f}x  {mode}; work items redone {st[1]} of {st[2]}, rows "
                  f"{st[5]}; est forced: rows redone {rows_forced} of {H * NQ}, outlying-norm keys per head {min(hi)}..{max(hi)}")


if __name__ == "__main__":
    main()
