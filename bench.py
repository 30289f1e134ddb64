#!/usr/bin/env python3
"""bench.py -- views/sec of the IGGT forward (N-view 518x518) on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W                      (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N ranks, RCCL)
  python bench.py --gpus N ...                                       (no launcher: spawns the N ranks itself, same thing)

Workload (BASELINE.json configs[2] / [3]): 32 synthetic views @ 518x518, end-to-end forward =
DINOv2 backbone + 24 x (frame, global) blocks + camera head + depth head + point head, seeded synthetic
weights of the reference architecture (no checkpoint reachable; iggt_official_amd/synthetic.py, the same
(weights, images) the reference fixture tests/golden/full_s32_518_stress.pt was produced with, so the outputs of the
TIMED model are compared with the reference's -- `output_check` below), 16-bit MFMA operands with fp32 accumulation:
fp16 by default -- the format whose outputs stay within 1e-3 of the reference's fp32 CPU path (tests/test_e2e_gpu.py);
IGGT_OPERAND_DTYPE=bf16 selects the reference's own GPU precision (demo.py:193-195), same MFMA rate.  518 is not a multiple of 28, where
the reference's part head raises, so the step produces the geometry outputs --
exactly what the reference can produce at this size.  One "step" = one full forward of all views.
Views are sharded over ranks ("strong" scaling: 32 views total for every N); the only data-path
collective is the K/V all-gather in front of each global attention (iggt_official_amd/dist.py).

Prints ONE JSON line (rank 0) with value = total views / second, plus
  "roofline": global-attention flash kernel, algorithmic FLOPs (4*Nq*Nk*C per launch) / mean launch
              duration measured with HIP events on the launch stream inside the timed region, against
              the 2.5 PFLOP/s dense 16-bit (bf16 = fp16) MFMA peak;
  "output_check": relative l2 / max errors of the timed model's outputs against strided samples of the REFERENCE
              outputs (fixture produced by the reference modules on CPU fp32 at this very configuration) plus a
              checksum (mean / abs-sum per output) that tests/test_headline_gpu.py pins through the same fixture;
  "precision_rung": which transformer blocks run on the x3 rung (fp16 hi + lo operand pairs, three MFMA passes per product;
              iggt_official_amd/precision.py) on THIS checkpoint -- none on the BASELINE one -- and, when any do, the step time and the
              output check of one extra forward with the rung switched off.  `--weights MODE` selects a heavy-tailed synthetic
              checkpoint (with --views 8 the outputs are checked against the matching dose fixture);
  "full_model": the WHOLE model incl. the instance-feature branch (`part_feat`) at the same view count @ 532 x 532 (where the
              reference's part head is defined): views/s, the global-attention roofline at that size, per-kernel-family entries
              for the part branch (its convolutions, the two window-attention stages, the token cross-attention) and the
              output check -- part_feat included -- against the reference fixture of that configuration;
  "cpu_baseline": the CPU restatement of the reference (oracle/restate.py, kind "port") timed on this
              box's host cores on a bounded sample (4 views @ 518x518 = BASELINE.json configs[0]'s size), plus a
              32-view figure extrapolated from a row-sampled global attention (flagged as such), rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak
# What the matrix pipe sustains from registers alone once the operands are real data (the chip is power-limited: 2.38 GHz
# with zero / constant operands, 1.51 GHz fp16 / 1.64 GHz bf16 with random ones) -- probes/mfma_power.hip,
# profiles/r02_mfma_power.txt.  Reported beside the contract's peak, never instead of it.
MFMA_SUSTAINED_REAL_DATA_TFLOPS = {"f16": 1581.0, "bf16": 1721.0}
# HBM-side bytes per global-attention launch, from rocprofv3 PMC passes on the kernel named in the entry
# (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes; MI355X_MICROARCH.md section HBM).  PMC counters cannot
# be read live inside the timed run, so the figure is keyed by (views, size, n_gpus, kernel label) and reported as
# null whenever the kernel actually launched is not the one it was measured on.
PMC_TRAFFIC = {}
try:
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "attn_traffic.json")) as _f:
        PMC_TRAFFIC = json.load(_f)
except Exception:  # noqa: BLE001
    PMC_TRAFFIC = {}
FIXTURES = {(32, 518): ("full_s32_518_stress", 8), (8, 518): ("full_s8_518_stress", 7),    # (views, size) -> (fixture, image seed)
            (32, 532): ("full_s32_532_stress", 11), (8, 532): ("full_s8_532_stress", 10)}    # 532^2: the WHOLE model incl. part_feat
FULL_MODEL_SIZE = 532     # nearest size above 518 on which the reference's part head is defined (H, W in 28 N)
# --weights <mode>: the heavy-tailed dose fixtures (8 views @ 518^2; iggt_official_amd/synthetic.py "trained_like", oracle/make_golden.py)
DOSE_FIXTURES = {"trained_like(qk=0.5,norm=0.5)": "full_s8_518_tlA", "trained_like(qk=0.75,norm=0.5)": "full_s8_518_tlB",
                 "trained_like": "full_s8_518_tlC", "trained_like(qk=1,norm=0.5)": "full_s8_518_tlD",
                 "trained_like(qk=0.75,norm=0.75)": "full_s8_518_tlE", "trained_like(qk=0,norm=1)": "full_s8_518_tlF"}


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:  # noqa: BLE001
        pass
    return n


def _cpu_baseline_worker(sample_views, size, cores):
    """(child process) oracle port on the host cores: one `sample_views`-view geometry forward, then the global
    attention of ONE view's queries against the keys of `sample_views` and of 32 views (row-sampled: full K/V, 1374
    query rows) for the extrapolation to the bench configuration."""
    import torch.nn.functional as F

    from oracle import restate, weights

    torch.set_num_threads(cores)
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")) as f:
        schema = json.load(f)
    sd = weights.fill_state_dict(schema, seed=0, mode="default")
    images = weights.make_images(sample_views, size, size, seed=0)
    with torch.no_grad():      # warm-up (thread pool, allocator, oneDNN primitive caches): one view at a quarter of the area
        restate.iggt_forward(sd, weights.make_images(1, 266, 266, seed=0), with_part=False)
    t0 = time.perf_counter()
    with torch.no_grad():
        restate.iggt_forward(sd, images, with_part=False)
    dt = time.perf_counter() - t0
    P = 5 + (size // 14) ** 2
    att = {}
    with torch.no_grad():
        for views in (sample_views, 32):
            g = torch.Generator().manual_seed(views)
            q = torch.randn(1, 16, P, 64, generator=g)
            k = torch.randn(1, 16, views * P, 64, generator=g)
            v = torch.randn(1, 16, views * P, 64, generator=g)
            F.scaled_dot_product_attention(q[:, :, :64], k, v)   # warm-up
            t0 = time.perf_counter()
            F.scaled_dot_product_attention(q, k, v)
            att[views] = time.perf_counter() - t0
    print(json.dumps({"dt": dt, "attn_rows": att}), flush=True)


def cpu_baseline(sample_views, size, budget_s=240):
    """Runs the oracle (kind "port") in a child process with a wall-clock bound so that a slow or
    throttled host can never stall the benchmark."""
    import subprocess

    cores = usable_cores()
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(sample_views), str(size),
           str(cores)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s, cwd=ROOT)
        rec = json.loads(out.stdout.strip().splitlines()[-1])
        dt = rec["dt"]
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "views/s", "cores": cores, "kind": "port",
                "sample": f"{sample_views} view(s) @ {size}x{size} did not finish within {budget_s}s on {cores} cores"}
    res = {"value": sample_views / dt, "unit": "views/s", "cores": cores, "kind": "port",
           # A Python reference cannot travel to the GPU box in any form, so `kind` stays "port".  What the substitution is worth was
           # measured where both exist (build container, 8 cores, 4 views @ 518^2, one same-size warm-up, one timed forward each;
           # probes/cpu_reference_vs_port.py -> profiles/r06_cpu_reference_vs_port.txt): reference modules 26.28 s, port 25.35 s,
           # outputs bit-identical (same ATen kernels in the same order; the reference also keeps 24 instead of 4 token layers).
           "port_vs_reference": {"speed_ratio": 1.04, "reference_s": 26.28, "port_s": 25.35, "outputs": "bit-identical",
                                 "where": "build container (8 cores), 4 views @ 518x518, same protocol for both",
                                 "source": "profiles/r06_cpu_reference_vs_port.txt"},
           "sample": f"{sample_views} view(s) @ {size}x{size}, full geometry forward (DINOv2 + 24x(frame,global) + "
                     f"camera/depth/point heads) of oracle/restate.py (a PORT of the reference, not the reference "
                     f"itself: /root/reference does not exist on the GPU box), fp32 torch CPU, {cores} threads, one timed run "
                     f"{dt:.1f}s after a one-view warm-up"}
    a = {int(k): v for k, v in rec.get("attn_rows", {}).items()}
    if sample_views in a and 32 in a and sample_views != 32:
        # per-view time at 32 views = measured per-view time + 24 blocks x (one view's global-attention rows against
        # 32 views of keys - the same against `sample_views` views of keys); everything else is linear in the views
        per_view_32 = dt / sample_views + 24.0 * (a[32] - a[sample_views])
        res["extrapolated_32_views"] = {
            "value": 1.0 / per_view_32, "unit": "views/s", "extrapolated": True,
            "how": f"per-view time of the {sample_views}-view run + 24 x (row-sampled global attention: 1374 query rows x "
                   f"16 heads against 32 views of keys {a[32]:.2f}s - against {sample_views} views {a[sample_views]:.2f}s)"}
    return res


def output_check(pred, S, H, v0, v1, dev, mode="stress"):
    """Compare the timed model's outputs (this rank's views v0:v1) with the reference fixture of this configuration
    (tests/golden/<fixture>.pt: strided samples written by the REFERENCE modules on CPU fp32, oracle/make_golden.py).
    Returns None when no fixture exists for (S, H)."""
    fx = FIXTURES.get((S, H)) if mode == "stress" else ((DOSE_FIXTURES[mode], 7) if (mode in DOSE_FIXTURES and (S, H) == (8, 518)) else None)
    path = os.path.join(ROOT, "tests", "golden", (fx[0] if fx else "-") + ".pt")
    if fx is None or not os.path.exists(path):
        return None
    g = torch.load(path, map_location="cpu", weights_only=False)
    ss = g["meta"]["spatial_stride"]
    out = {"fixture": "tests/golden/" + fx[0] + ".pt", "errors": {}, "checksum": {}}
    for k in ("depth", "depth_conf", "world_points", "world_points_conf") + (("part_feat",) if ("part_feat" in g and "part_feat" in pred) else ()):
        got = (pred[k][:, :, :, ::ss, ::ss] if k == "part_feat" else pred[k][:, :, ::ss, ::ss]).double().cpu()
        ref = g[k][:, v0:v1].double()
        d = got - ref
        out["errors"][k] = {"l2": float(d.norm() / ref.norm()), "max": float(d.abs().max() / ref.abs().max())}
        v = pred[k].double()
        out["checksum"][k] = {"mean": float(v.mean()), "abs_sum": float(v.abs().sum())}
        if v0 == 0 and v1 == S:   # whole tensor on this rank: the reference's whole-tensor statistics apply
            st = g["stats"][k]
            out["checksum"][k]["ref_mean"] = st["mean"]
            out["checksum"][k]["ref_abs_sum"] = st["abs_sum"]
    pe = torch.stack(pred["pose_enc"], 0).double().cpu()
    ref = g["pose_enc"].double()
    out["errors"]["pose_enc"] = {"l2": float((pe - ref).norm() / ref.norm()),
                                 "max": float((pe - ref).abs().max() / ref.abs().max())}
    out["max_l2"] = max(e["l2"] for e in out["errors"].values())
    out["tolerance"] = 1e-3
    out["ok"] = bool(out["max_l2"] < 1e-3)
    return out


def _secondary_rooflines(g_recs, c_recs):
    """[{gemm}, {conv}]: algorithmic FLOPs / HIP-event time per kernel family, against the same dense 16-bit MFMA peak.  The
    head convolutions compute fp32-grade products out of three bf16 MFMAs each (split-bf16): their
    ALGORITHMIC rate is what is reported; `mfma_equivalent` is 3x that."""
    out = []
    by = {}
    for ms, (name, M, N, K), _tag in g_recs:
        d = by.setdefault(name, [0.0, 0.0, 0, (M, N, K)])
        d[0] += ms
        d[1] += 2.0 * M * N * K
        d[2] += 1
    if by:
        tot_ms = sum(d[0] for d in by.values())
        tot_fl = sum(d[1] for d in by.values())
        out.append({"kernel": "gemm_bf16_t256pp_kernel / gemm_bf16_dma_kernel (trunk Linears incl. bias / GELU / LayerScale / "
                              "residual epilogues)", "bound": "mfma", "achieved": tot_fl / (tot_ms * 1e-3) / 1e12,
                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tot_fl / (tot_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                    "ms_per_forward": tot_ms, "launches": sum(d[2] for d in by.values()),
                    "per_shape": {k: {"M_N_K": list(d[3]), "launches": d[2], "ms_mean": d[0] / d[2],
                                      "tflops": d[1] / (d[0] * 1e-3) / 1e12} for k, d in by.items()}})
    if c_recs:
        ms = sum(r[0] for r in c_recs)
        fl = sum(r[1][0] for r in c_recs)
        issued = sum(r[1][0] * r[1][1] for r in c_recs)     # MFMA work the kernels issue: 2 or 3 passes per product
        two = sum(1 for r in c_recs if r[1][1] == 2)
        out.append({"kernel": "conv3x3_halo_kernel / conv_igemm_kernel (DPT head convolutions: exact fp16 hi + lo activations x fp16 "
                              "weights with mean-input compensation = 2 MFMA passes on the large layers, split-bf16 = 3 on the rest)",
                    "bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "mfma_equivalent_tflops": issued / (ms * 1e-3) / 1e12,
                    "ms_per_forward": ms, "launches": len(c_recs), "launches_two_pass": two,
                    "note": "timed with the two DPT heads in line (in the timed region the depth head runs beside the point head on a "
                            "second stream)"})
    return out


def _worstcase_attention(S, P, dev, dt):
    """`roofline_worstcase`: the global-attention launch of this configuration (Nq = Nk = S * P, 16 heads, the product's adaptive
    static-bound dispatcher with its estimated-shift workspace, steady state) on four synthetic score regimes -- per-head
    LayerNorm of noise (what the synthetic checkpoint produces), trained-like q/k-norm affines (log-normal per-channel scales),
    8 sink keys of 10x the norm, register-token query rows of 30x the norm (probes/attn_static_robustness.py) -- HIP events, the
    slowest regime reported against the same peak.  The headline `roofline` entry is measured inside the model on the synthetic
    checkpoint, i.e. in the first regime."""
    from iggt_official_amd import _C

    H, C, T = 16, 1024, S * P
    out = {}
    for kind in ("noise", "affine", "sinks", "registers"):
        g = torch.Generator(device=dev).manual_seed(3)
        x = torch.randn(T, 3, H, 64, generator=g, device=dev)
        qk = x[:, :2]
        qk = (qk - qk.mean(-1, keepdim=True)) / qk.std(-1, keepdim=True, unbiased=False)
        if kind != "noise":
            gam = torch.exp(torch.randn(2, H, 64, generator=g, device=dev))
            qk = qk * (gam / gam.pow(2).mean(-1, keepdim=True).sqrt())[None]
        x[:, :2] = qk
        x[:, 0] *= 0.125 * _C.LOG2E * 1.3
        if kind == "sinks":
            x[torch.randperm(T, generator=g, device=dev)[:8], 1] *= 10.0
        if kind == "registers":
            x[(torch.arange(S, device=dev)[:, None] * P + torch.arange(5, device=dev)[None]).reshape(-1), 0] *= 30.0
        qkv = x.reshape(T, 3 * C).to(dt)
        del x, qk
        qkmax = torch.zeros(_C.QKMAX_NUMEL, device=dev)
        _C.k_rownorm_max(qkv[:, C:2 * C], qkmax)
        o = torch.empty(T, C, dtype=dt, device=dev)
        flags = torch.zeros(H * ((T + 127) // 128), dtype=torch.int32, device=dev)
        est_ws = torch.zeros(_C.static_attn_est_ws_bytes(1, H, T, T), dtype=torch.uint8, device=dev)
        guard = _C.new_attn_guard(dev)

        def launch():
            _C.flash_attn_d64_static(qkv, qkv[:, C:], qkv[:, 2 * C:], o, 1, H, T, T, 0, 3 * C, 0, 3 * C, 0, 3 * C, 0, C, qkmax,
                                     flags, 0, None, guard, None, est_ws=est_ws, key_period=P, key_nspecial=5)

        for _ in range(3):          # the switch settles within two calls (norm bound -> estimated shift)
            launch()
        torch.cuda.synchronize()
        ms = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        st = guard.tolist()
        out[kind] = {"ms_per_launch": sum(ms) / len(ms),
                     "mode": "online-max only" if (st[0] > 0 or st[1] < 0) else ("estimated shift" if st[4] == 1 else "norm bound"),
                     "work_items_redone": st[1], "rows_handed_over": st[5]}
        del qkv, o, est_ws
    worst = max(out, key=lambda k: out[k]["ms_per_launch"])
    flops = 4.0 * T * T * C
    ach = flops / (out[worst]["ms_per_launch"] * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "global-attention launch (adaptive static-bound dispatcher: norm bound / estimated shift / "
                                       "online-max) on four synthetic score regimes, slowest reported", "regime": worst,
            "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
            "ms_per_launch": out[worst]["ms_per_launch"], "flops_per_launch": flops, "per_regime": out,
            "timing_note": "HIP events, mean of 10 launches after 3 warm-up launches of a cold call site"}


def _full_model_leg(model, dev, S, mode, steps):
    """`full_model`: the WHOLE model -- trunk, camera / depth / point heads AND the instance-feature branch (`part_feat`:
    SamProjector + PartHead with its token cross-attention and the two window-attention stages; reference vggt.py:204-218) --
    at S views @ 532 x 532, the nearest size above BASELINE's 518 on which the reference's part head is defined (H, W in 28 N).
    The headline configuration cannot carry it: at 518^2 the reference raises inside the part head.  Same
    model object as the headline (same weights, same packs), synthetic images of the reference fixture of this size; timed like the
    headline (synchronise, `steps` forwards, synchronise), outputs incl. part_feat checked against the fixture the REFERENCE
    modules produced at this very configuration, then ONE extra eager forward with HIP events around every launch of the part
    branch's kernel families (the two DPT heads in line for it)."""
    from iggt_official_amd import _C, precision, profiling, synthetic
    from iggt_official_amd.models import vggt as _mv

    H = FULL_MODEL_SIZE
    P, C = 5 + (H // 14) ** 2, 1024
    fx = FIXTURES.get((S, H))
    images = synthetic.make_images(S, H, H, seed=fx[1] if fx else 1234, device=dev)
    torch.cuda.reset_peak_memory_stats(dev)
    for _ in range(2):                # warm-up: workspaces of the larger grid, the part branch's weight packs
        model(images)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = model(images)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    assert "part_feat" in out and all(torch.isfinite(v).all() for v in out.values() if torch.is_tensor(v))
    check = output_check(out, S, H, 0, S, dev, mode)
    names = ("global_attn", "conv", "window_attn", "cross_attn", "part_branch")
    orig = _mv._HEAD_STREAMS
    try:
        _mv._HEAD_STREAMS = "0"
        for nm in names:
            profiling.enable(nm)
        model(images)
        torch.cuda.synchronize()
        recs = {nm: profiling.summarize(profiling.disable(nm) or []) for nm in names}
    finally:
        _mv._HEAD_STREAMS = orig
        for nm in names:
            profiling.disable(nm)

    def fam(rs, peak, unit_flops=True):
        t = sum(r[0] for r in rs)
        fl = sum(r[1][1] if isinstance(r[1][0], str) else r[1][0] for r in rs)
        return {"launches": len(rs), "ms_per_forward": t, "algorithmic_tflop_per_forward": fl / 1e12,
                "achieved": fl / max(t, 1e-9) / 1e9, "peak": peak, "unit": "TFLOP/s", "frac": fl / max(t, 1e-9) / 1e9 / peak}

    ga = recs["global_attn"]
    ga_ms = sum(r[0] for r in ga) / max(len(ga), 1)
    ga_fl = 4.0 * (S * P) ** 2 * C
    part_conv = [r for r in recs["conv"] if r[2] == "part"]
    dpt_conv = [r for r in recs["conv"] if r[2] != "part"]
    sec = []
    if part_conv:
        e = fam(part_conv, MFMA_BF16_PEAK_TFLOPS)
        passes = sum(r[1][0] * r[1][1] for r in part_conv) / max(sum(r[1][0] for r in part_conv), 1.0)
        e.update(kernel="conv_igemm_kernel / conv3x3_halo_kernel, part branch (SamProjector `Projects` stacks with folded BatchNorm, "
                        "PartHead fusion blocks, SwinSA / SwinCA convolutions, the part head's Linear layers as 1x1 convolutions)",
                 bound="mfma", mfma_passes_per_product=passes, mfma_equivalent_tflops=e["achieved"] * passes)
        sec.append(e)
    for kind in sorted({r[1][0] for r in recs["window_attn"]}):
        rs = [r for r in recs["window_attn"] if r[1][0] == kind]
        e = fam(rs, 157.3)
        nbytes, t = sum(r[1][2] for r in rs), sum(r[0] for r in rs)
        e.update(kernel=f"window_attn_kernel ({kind}; exact fp32 MFMA v_mfma_f32_32x32x2_f32, two waves per (window, head), K / V tiles in LDS)",
                 bound="hbm" if "HAB" in kind else "mfma-fp32",
                 hbm={"algorithmic_bytes_per_forward": nbytes, "achieved": nbytes / max(t, 1e-9) / 1e6, "peak": 8000.0, "unit": "GB/s",
                      "frac": nbytes / max(t, 1e-9) / 1e6 / 8000.0},
                 peak_note="TFLOP/s against the fp32 matrix peak of the guide (157.3); the counters (profiles/r06_window_attn_pmc.txt) "
                           "show HBM traffic = the algorithmic bytes: HAB moves 6.1 GB per 32-view pass and is HBM-bound, OCAB re-reads "
                           "k / v 2.25x (overlapping 12 x 12 windows) and keeps the matrix pipe 64 % busy")
        sec.append(e)
    if recs["cross_attn"]:
        e = fam(recs["cross_attn"], 157.3)
        e.update(kernel="attn_f32_kernel<32> (PartHead.cross_attention_2: g^2 part tokens x g^2 point-feature tokens, 8 heads, "
                        "exact fp32 MFMA v_mfma_f32_32x32x2_f32)", bound="mfma-fp32")
        sec.append(e)
    if dpt_conv:
        e = fam(dpt_conv, MFMA_BF16_PEAK_TFLOPS)
        e.update(kernel="conv3x3_halo_kernel / conv_igemm_kernel, the two DPT heads at this size", bound="mfma")
        sec.append(e)
    pb = sum(r[0] for r in recs["part_branch"])
    return {
        "workload": f"{S} views @ {H}x{H}, IGGT forward incl. part_feat (DINOv2 + 24x(frame,global) + camera / depth / point heads + "
                    "SamProjector + PartHead), same synthetic checkpoint as the headline",
        "value": S / (ms * 1e-3), "unit": "views/s", "ms_per_step": ms, "steps": steps, "warmup": 2, "views": S, "image_size": H,
        "tokens_per_view": P, "peak_memory_gib": torch.cuda.max_memory_allocated(dev) / 2.0 ** 30,
        "part_branch_ms_per_forward": pb,
        "part_branch_note": "HIP events around part_adaptor + part_head in the per-kernel forward (heads in line); in the timed "
                            "steps the depth head runs beside the point head on a second stream",
        "roofline": {"bound": "mfma", "kernel": _C.attn_kernel_label(1, 16, S * P, S * P, precision.operand_name(),
                                                                     static_bound=precision.static_softmax(), with_part_ws=True)
                     + " (global attention)", "achieved": ga_fl / max(ga_ms, 1e-9) / 1e9, "peak": MFMA_BF16_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": ga_fl / max(ga_ms, 1e-9) / 1e9 / MFMA_BF16_PEAK_TFLOPS, "ms_per_launch": ga_ms,
                     "launches_timed": len(ga), "flops_per_launch": ga_fl},
        "roofline_secondary": sec,
        "output_check": check,
    }


def _self_launch(n):
    """Re-execute this script as n ranks of one node: python -m torch.distributed.run --nnodes=1 --nproc-per-node n
    --master-addr 127.0.0.1 --master-port <free port> bench.py <the same arguments>.  stdout / stderr pass through."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    rc = subprocess.call(cmd, env=env, cwd=ROOT)
    if rc != 0:
        raise SystemExit(rc)


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-baseline-worker":
        return _cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=32)
    ap.add_argument("--size", type=int, default=518)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-model", action="store_true",
                    help="skip the `full_model` leg (the whole model incl. part_feat at 532^2, a few extra forwards behind the timed region)")
    ap.add_argument("--cpu-sample-views", type=int, default=4)
    ap.add_argument("--graphs", choices=["auto", "on", "off"], default="auto",
                    help="replay the forward as hipGraph segments (auto: when N > 1, where the per-rank forward is host-bound)")
    ap.add_argument("--emulate-world", type=int, default=0, metavar="W",
                    help="developer mode, --gpus 1 only: time what ONE (middle) rank of a W-GPU run computes -- its S / W views, "
                         "the global attention over the keys of all S views (gathers replaced by local copies, "
                         "probes/emulated_shard.py).  The outputs are not the model's; no output check")
    ap.add_argument("--weights", default="stress", metavar="MODE",
                    help="synthetic checkpoint mode (iggt_official_amd/synthetic.py): 'stress' (default, the BASELINE configuration) "
                         "or a heavy-tailed dose such as 'trained_like(qk=1,norm=0.5)' -- what the x3 precision rung costs where it "
                         "engages (`precision_rung` in the JSON line); with --views 8 the outputs are checked against the dose fixture")
    ap.add_argument("--random-init", action="store_true",
                    help="torch random-init weights and images instead of the synthetic checkpoint (no output check)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` typed like the N = 1 case: spawn the ranks ourselves (one process per GPU under
        # torch.distributed.run, the launcher the driver uses) and relay rank 0's JSON line
        return _self_launch(args.gpus)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world} of the launcher")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback for the product path)")
    # developer switch (not a product path): IGGT_BENCH_SINGLE_DEVICE=1 runs all ranks on cuda:0 over gloo, so that the
    # multi-rank control flow of this script can be exercised on a one-GPU box (RCCL refuses two ranks on one device)
    single_dev = os.environ.get("IGGT_BENCH_SINGLE_DEVICE", "0") == "1"
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    shard = None
    if world > 1 and single_dev:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    # developer switch: IGGT_FORCE_COLLECTIVES=1 at N = 1 runs the sharded code path with its real RCCL calls in a world of
    # one rank (everything but transport between devices): what the collectives cost per forward on this hardware
    force_coll = world == 1 and os.environ.get("IGGT_FORCE_COLLECTIVES", "0") == "1"
    if force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    from iggt.models.vggt import IGGT
    from iggt_official_amd import _C, precision, profiling, synthetic
    from iggt_official_amd.dist import ViewShard, view_partition

    _C.load()
    S, H = args.views, args.size
    if S % world:
        raise SystemExit(f"--views {S} must be divisible by the number of GPUs {world}")
    torch.manual_seed(0)  # identical weights on every rank
    with torch.device(dev):
        model = IGGT(part_on_invalid_grid="skip").eval()
    emu = args.emulate_world if (args.emulate_world > 1 and world == 1) else 0
    if emu and S % emu:
        raise SystemExit(f"--views {S} must be divisible by --emulate-world {emu}")
    if world > 1 or force_coll:
        shard = ViewShard()
        model.set_view_shard(shard)
    elif emu:
        from probes.emulated_shard import EmulatedShard   # developer tool: stands in for the other ranks on one GPU

        shard = EmulatedShard(emu)
        model.set_view_shard(shard)
    v0, v1 = shard.local_views(S) if emu else view_partition(S, world, rank)
    if args.random_init:
        g = torch.Generator(device="cpu").manual_seed(1234)
        images = torch.rand(S, 3, H, H, generator=g)[v0:v1].to(dev)
        data = "synthetic (torch random-init weights, uniform random images)"
    else:
        # the synthetic checkpoint + images of the reference fixtures (hash-generated on the device, bit-identical to
        # the CPU values the reference was run with): the timed outputs can be checked against the reference
        with open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")) as f:
            schema = json.load(f)
        sd = synthetic.fill_state_dict(schema, seed=0, mode=args.weights, device=dev)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        del sd
        assert not [u for u in unexpected if not u.startswith("track_head.")], unexpected
        iseed = FIXTURES.get((S, H), (None, 1234))[1]
        images = synthetic.make_images(S, H, H, seed=iseed, device=dev)[v0:v1].contiguous()
        data = f"synthetic (seeded hash weights '{args.weights}' seed 0 and hash-noise images, iggt_official_amd/synthetic.py)"

    graphs = args.graphs == "on" or (args.graphs == "auto" and (world > 1 or emu))
    if graphs:
        model.enable_graphs(True)   # first call captures (inside the warm-up)

    def step():
        return model(images)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    graph_note = None
    if graphs:
        # the first call captures.  If capture fails (deterministically, so on every rank at the same point) fall back to
        # eager launches rather than lose the measurement; the JSON line says which mode was timed.
        try:
            step()
            fence()
        except Exception as ex:  # noqa: BLE001
            graph_note = f"hipGraph capture failed, timed eager launches instead: {ex!r}"[:300]
            graphs = False
            model.enable_graphs(False)
            torch.cuda.synchronize()
    for _ in range(max(args.warmup - (1 if graphs else 0), 0)):
        step()
    fence()
    if not graphs:
        profiling.enable("global_attn")
        profiling.enable("global_attn_x3")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if graphs:
        # HIP events cannot be recorded inside a replayed graph: the attention launches are timed in ONE extra eager
        # forward right after the timed region (same kernels, same buffers); `value` comes from the graph replays
        out = {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in out.items()}
        model.enable_graphs(False)
        profiling.enable("global_attn")
        profiling.enable("global_attn_x3")
        step()
        fence()
    recs = profiling.summarize(profiling.disable("global_attn"))
    recs_x3 = profiling.summarize(profiling.disable("global_attn_x3") or [])
    # the precision rung (precision.py "x3"): which blocks run on fp16 operand pairs, and -- one extra forward behind the timed
    # region with the rung switched off -- what that costs on this checkpoint (nothing on the BASELINE one: no block escalates)
    rung = model.aggregator.escalation_report()
    rung = {"blocks": rung["blocks"], "escalated": len(rung["x3"]), "ill_conditioned_by_own_figures": len(rung["own_verdict"]),
            "escalated_blocks": rung["x3"] if len(rung["x3"]) < rung["blocks"] else "all", "bf16_fallback": rung["bf16_fallback"],
            "min_participation_ratio": rung["min_participation_ratio"], "max_logit_rms": rung["max_logit_rms"],
            "min_participation_ratio_untrimmed": rung["min_participation_ratio_untrimmed"],
            "thresholds": {"participation_ratio_below": precision.ESC_PR_MIN, "channels_trimmed_per_1024": precision.ESC_PR_TRIM,
                           "logit_rms_above": precision.ESC_LOGIT_RMS_MAX},
            "policy": precision.escalation()}
    if rung["escalated"]:
        was_graphs = model._graphs_on
        prev_policy = precision.escalation_policy()      # "auto", or IGGT_ESCALATE=all (the maximum-parity mode): restored below
        try:
            model.enable_graphs(False)
            precision.set_escalation("off")
            step()
            fence()
            t1 = time.perf_counter()
            step()
            fence()
            rung["ms_per_step_single_fp16_operands"] = (time.perf_counter() - t1) * 1e3
            rung["single_fp16_output_check"] = None if (args.random_init or emu) else output_check(step(), S, H, v0, v1, dev, args.weights)
        finally:
            precision.set_escalation(prev_policy)
            step()            # re-pack on the rung (the legs below time the shipping configuration)
            fence()
            model.enable_graphs(was_graphs)
    soft = model.aggregator.static_softmax_stats() if precision.static_softmax() else None
    # secondary legs, each ONE extra eager forward behind the timed region (the timed steps carry no extra events):
    #   GEMMs (qkv / proj / fc1 / fc2 of the 72 blocks) and head convolutions, HIP events per launch;
    #   the same global-attention kernel with bf16 operands (north_star's named operand type, the reference's autocast mode)
    secondary, bf16_leg = None, None
    orig_dt = precision.operand_dtype()
    from iggt_official_amd.models import vggt as _mv
    orig_hs = _mv._HEAD_STREAMS
    try:
        if graphs:
            model.enable_graphs(False)
        # per-launch events only mean something when the launches do not share the chip: the depth head normally runs beside the
        # point head on a second stream (models/vggt.py); for this leg the two heads run one after the other
        _mv._HEAD_STREAMS = "0"
        profiling.enable("gemm"), profiling.enable("conv")
        step()
        fence()
        g_recs = profiling.summarize(profiling.disable("gemm"))
        c_recs = profiling.summarize(profiling.disable("conv"))
        _mv._HEAD_STREAMS = orig_hs
        secondary = _secondary_rooflines(g_recs, c_recs)
        if precision.operand_name() != "bf16" and os.environ.get("IGGT_BENCH_BF16_LEG", "1") != "0":
            precision.set_operand_dtype("bf16")
            step()
            fence()
            profiling.enable("global_attn")
            step()
            fence()
            b_recs = profiling.summarize(profiling.disable("global_attn"))
            if b_recs:
                bf16_leg = sum(r[0] for r in b_recs) / len(b_recs)
    except Exception as ex:  # noqa: BLE001  (never lose the main line to a side measurement)
        secondary = secondary or [{"error": repr(ex)[:300]}]
    finally:
        _mv._HEAD_STREAMS = orig_hs
        for nm in ("gemm", "conv"):
            profiling.disable(nm)
        precision.set_operand_dtype(orig_dt)
    worst = None
    if world == 1 and not emu and os.environ.get("IGGT_BENCH_WORSTCASE", "1") != "0" and precision.static_softmax():
        try:
            worst = _worstcase_attention(S, 5 + (H // 14) ** 2, dev, precision.operand_dtype())
        except Exception as ex:  # noqa: BLE001  (never lose the main line to a side measurement)
            worst = {"error": repr(ex)[:300]}
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert all(torch.isfinite(v).all() for v in out.values() if torch.is_tensor(v))
    check = None if (args.random_init or emu) else output_check(out, S, H, v0, v1, dev, args.weights)
    peak_mem = torch.cuda.max_memory_allocated(dev) / 2.0 ** 30
    full_model = None
    if world == 1 and not emu and not force_coll and not args.no_full_model and not args.random_init and H == 518:
        try:
            if graphs:
                model.enable_graphs(False)
            full_model = _full_model_leg(model, dev, S, args.weights, max(1, min(args.steps, 5)))
        except Exception as ex:  # noqa: BLE001  (never lose the main line to a side measurement)
            full_model = {"error": repr(ex)[:300]}
    if world > 1 and check is not None:   # worst rank decides
        t = torch.tensor([check["max_l2"]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        check["max_l2_all_ranks"] = float(t.item())
        check["ok"] = bool(check["max_l2_all_ranks"] < check["tolerance"])

    if rank == 0:
        P = 5 + (H // 14) ** 2
        C = 1024
        Nq, Nk = (S // (emu or world)) * P, S * P
        x3_attn = not recs and bool(recs_x3)     # every global block on the x3 rung: its attention kernel is the dominant one
        if x3_attn:
            recs = recs_x3
        ms = sum(r[0] for r in recs) / max(len(recs), 1)
        flops = 4.0 * Nq * Nk * C
        achieved = flops / (max(ms, 1e-9) * 1e-3) / 1e12
        if (world > 1 or emu) and precision.static_softmax():
            kernel_label = (f"flash_attn_d64_v3_kernel<QB=2,KVM=2,{precision.operand_name()},static-bound> own keys + "
                            f"{(emu or world) - 1} key segments + attn_combine_kernel")
        else:
            kernel_label = _C.attn_kernel_label(1, 16, Nq, Nk, precision.operand_name(), static_bound=precision.static_softmax(),
                                                with_part_ws=True)
        if x3_attn:
            kernel_label = "flash_attn_x3_kernel<f16 hi + lo pairs, 3 MFMA passes per product, online max>"
        traffic = PMC_TRAFFIC.get(f"{S}x{H}x{emu or world}:{kernel_label}", {})
        line = {
            "metric": "views/sec (N-view 518^2 forward)",
            "value": (v1 - v0 if emu else S) * args.steps / dt,
            "unit": "views/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": precision.operand_name(),
            "data": data,
            "graphs": bool(graphs),
            "peak_memory_gib": peak_mem,
            **({"graphs_note": graph_note} if graph_note else {}),
            **({"emulated_rank": {"world": emu, "rank": shard.rank, "views_of_this_rank": v1 - v0,
                                  "job_views_per_s_if_transport_were_free": S * args.steps / dt,
                                  "note": "ONE GPU computing what a middle rank of a W-GPU run computes: per-rank shapes and byte "
                                          "counts, gathers = local copies on the compute stream, outputs not the model's (other "
                                          "ranks' keys are copies of this rank's); `value` = this rank's own views per second"}}
               if emu else {}),
            **({"collectives": "forced: RCCL (backend nccl) in a world of one rank -- every all-gather of the sharded path "
                               "is issued, nothing leaves the device"} if force_coll else {}),
            "config": {"workload": f"{S} views @ {H}x{H}, IGGT forward (DINOv2 + 24x(frame,global) + camera/depth/"
                                   "point heads), synthetic weights, views sharded " + f"{S // (emu or world)}/GPU"
                                   + (f" (rank {shard.rank} of {emu} EMULATED on one GPU)" if emu else ""),
                       "views": S, "image_size": H, "tokens_per_view": P,
                       "parallelism": f"view-shard x{emu} (emulated rank)" if emu else f"view-shard x{world}"},
            "roofline": {"bound": "mfma", "kernel": kernel_label + " (global attention)",
                         "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
                         "sustained_mfma_rate_on_random_operands": MFMA_SUSTAINED_REAL_DATA_TFLOPS.get(precision.operand_name()),
                         "frac_of_sustained": achieved / MFMA_SUSTAINED_REAL_DATA_TFLOPS.get(precision.operand_name(),
                                                                                             MFMA_BF16_PEAK_TFLOPS),
                         "sustained_note": "register-only MFMA loop, random operands: the chip's power limit for real data "
                                           "(profiles/r02_mfma_power.txt); 'peak' is reached with zero operands only",
                         "traffic": traffic.get("bytes_per_launch"),
                         "traffic_note": traffic.get("note", "no rocprofv3 PMC pass recorded for this kernel / shape "
                                                             "(profiles/attn_traffic.json)"),
                         "algorithmic_bytes_per_launch": 2.0 * (Nq + Nk) * C * 2,   # Q + O rows of this rank, K + V of all views
                         "launches_timed": len(recs), "ms_per_launch": ms,
                         "timing_note": ("HIP events around the launches of one eager forward right after the timed region "
                                         "(the timed steps replay hipGraph segments)") if graphs else
                                        "HIP events around every launch inside the timed region",
                         "timed_region": ("static-bound kernel + its gated online-max pass over flagged tiles (none on this "
                                          "input: ~10 us)") if precision.static_softmax() else "one kernel launch",
                         "flops_per_launch": flops},
        }
        rung["ms_per_step"] = line["ms_per_step"]
        if x3_attn:
            line["roofline"]["mfma_equivalent_tflops"] = 3.0 * achieved
            line["roofline"]["note"] = ("x3 precision rung: `achieved` counts the ALGORITHMIC 4 Nq Nk C once; the kernel issues three "
                                        "fp16 MFMA passes per product")
        elif recs_x3:
            ms3 = sum(r[0] for r in recs_x3) / len(recs_x3)
            rung["global_attention_x3"] = {"launches": len(recs_x3), "ms_per_launch": ms3,
                                           "algorithmic_tflops": flops / (ms3 * 1e-3) / 1e12}
        line["precision_rung"] = rung
        if soft is not None:
            line["static_softmax"] = dict(soft, note="query tiles of the last forward that the static-bound kernel handed to the "
                                          "online-max pass, per block kind (include/iggt_hip.h guard)")
        if secondary is not None:
            line["roofline_secondary"] = secondary
        if bf16_leg is not None:
            line["roofline_bf16"] = {"bound": "mfma", "kernel": _C.attn_kernel_label(1, 16, Nq, Nk, "bf16", static_bound=precision.static_softmax(), with_part_ws=True)
                                     + " (global attention, IGGT_OPERAND_DTYPE=bf16: one extra forward after the timed region)",
                                     "achieved": flops / (bf16_leg * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": flops / (bf16_leg * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "ms_per_launch": bf16_leg}
        if worst is not None:
            line["roofline_worstcase"] = worst
        if check is not None:
            line["output_check"] = check
        if full_model is not None:
            line["full_model"] = full_model
        if world == 1 and not emu and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.cpu_sample_views, H)
            except Exception as ex:  # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(line), flush=True)
    if world > 1 or force_coll:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
