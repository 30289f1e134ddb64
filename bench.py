#!/usr/bin/env python3
"""bench.py -- views/sec of the IGGT forward (N-view 518x518) on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W                      (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N ranks, RCCL)

Workload (BASELINE.json configs[2] / [3]): 32 synthetic views @ 518x518, end-to-end forward =
DINOv2 backbone + 24 x (frame, global) blocks + camera head + depth head + point head, random-init
weights of the reference architecture (no checkpoint reachable), 16-bit MFMA operands with fp32 accumulation:
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
  "cpu_baseline": the CPU restatement of the reference (oracle/restate.py, kind "port") timed on this
              box's host cores on a bounded sample (2 views @ 518x518), rank 0 / N=1 only.
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
# HBM-side bytes per global-attention launch at 32 views x 518^2 on one GPU, from rocprofv3 PMC passes
# (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE): profiles/r01_attn_hbm_pmc.txt.  Not measurable live.
PMC_TRAFFIC_BYTES = {(32, 518, 1): 1.17e9}


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
    """(child process) oracle port on the host cores: one `sample_views`-view geometry forward."""
    from oracle import restate, weights

    torch.set_num_threads(cores)
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")) as f:
        schema = json.load(f)
    sd = weights.fill_state_dict(schema, seed=0, mode="default")
    images = weights.make_images(sample_views, size, size, seed=0)
    t0 = time.perf_counter()
    restate.iggt_forward(sd, images, with_part=False)
    print(json.dumps({"dt": time.perf_counter() - t0}), flush=True)


def cpu_baseline(sample_views, size, budget_s=240):
    """Runs the oracle (kind "port") in a child process with a wall-clock bound so that a slow or
    throttled host can never stall the benchmark."""
    import subprocess

    cores = usable_cores()
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(sample_views), str(size),
           str(cores)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s, cwd=ROOT)
        dt = json.loads(out.stdout.strip().splitlines()[-1])["dt"]
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "views/s", "cores": cores, "kind": "port",
                "sample": f"{sample_views} view(s) @ {size}x{size} did not finish within {budget_s}s on {cores} cores"}
    return {"value": sample_views / dt, "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"{sample_views} view(s) @ {size}x{size}, full geometry forward (DINOv2 + 24x(frame,global) + "
                      f"camera/depth/point heads) of oracle/restate.py, fp32 torch CPU, {cores} threads, one run "
                      f"{dt:.1f}s; global attention is O(S^2), so the per-view CPU rate at 32 views is lower"}


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
    ap.add_argument("--cpu-sample-views", type=int, default=1)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    shard = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from iggt.models.vggt import IGGT
    from iggt_official_amd import _C, precision, profiling
    from iggt_official_amd.dist import ViewShard, view_partition

    _C.load()
    S, H = args.views, args.size
    if S % world:
        raise SystemExit(f"--views {S} must be divisible by the number of GPUs {world}")
    torch.manual_seed(0)  # identical random-init weights on every rank
    with torch.device(dev):
        model = IGGT(part_on_invalid_grid="skip").eval()
    if world > 1:
        shard = ViewShard()
        model.set_view_shard(shard)
    v0, v1 = view_partition(S, world, rank)
    g = torch.Generator(device="cpu").manual_seed(1234)
    images = torch.rand(S, 3, H, H, generator=g)[v0:v1].to(dev)

    def step():
        return model(images)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    profiling.enable("global_attn")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    recs = profiling.summarize(profiling.disable("global_attn"))
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert all(torch.isfinite(v).all() for v in out.values() if torch.is_tensor(v))

    if rank == 0:
        P = 5 + (H // 14) ** 2
        C = 1024
        Nq, Nk = (S // world) * P, S * P
        ms = sum(r[0] for r in recs) / max(len(recs), 1)
        flops = 4.0 * Nq * Nk * C
        achieved = flops / (ms * 1e-3) / 1e12
        line = {
            "metric": "views/sec (N-view 518^2 forward)",
            "value": S * args.steps / dt,
            "unit": "views/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": precision.operand_name(),
            "data": "synthetic",
            "config": {"workload": f"{S} views @ {H}x{H}, IGGT forward (DINOv2 + 24x(frame,global) + camera/depth/"
                                   "point heads), random-init weights, views sharded " + f"{S // world}/GPU",
                       "views": S, "image_size": H, "tokens_per_view": P, "parallelism": f"view-shard x{world}"},
            "roofline": {"bound": "mfma", "kernel": f"flash_attn_d64_v3_kernel<2,2,{precision.operand_name()}> (global attention)",
                         "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
                         "traffic": PMC_TRAFFIC_BYTES.get((S, H, world)),
                         "traffic_note": "bytes/launch from rocprofv3 PMC passes, profiles/r01_attn_hbm_pmc.txt",
                         "algorithmic_bytes_per_launch": 4.0 * Nk * C * 2,
                         "launches_timed": len(recs), "ms_per_launch": ms,
                         "flops_per_launch": flops},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.cpu_sample_views, H)
            except Exception as ex:  # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
