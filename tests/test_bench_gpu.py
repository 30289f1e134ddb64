"""GPU: the bench.py contract -- one JSON line with the roofline object, and the multi-rank control flow (two ranks on this
one GPU over gloo: IGGT_BENCH_SINGLE_DEVICE=1; RCCL refuses two ranks on one device) with hipGraph segments, the
max-over-ranks timing and the per-rank output check against the reference fixture."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(out):
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_single_gpu_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--views", "8", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["output_check"]["ok"]
    # round 5: the BASELINE checkpoint pays nothing for the precision rung
    assert d["precision_rung"]["escalated"] == 0 and d["precision_rung"]["blocks"] == 72 and d["precision_rung"]["policy"] == "auto"
    # round 6: the whole model incl. the instance-feature branch at 532^2 rides in the default line
    fm = d["full_model"]
    assert fm["image_size"] == 532 and fm["views"] == 8 and fm["value"] > 0 and fm["part_branch_ms_per_forward"] > 0
    assert fm["output_check"]["ok"] and fm["output_check"]["fixture"].endswith("full_s8_532_stress.pt")
    assert fm["output_check"]["errors"]["part_feat"]["l2"] < 1e-3
    kinds = " | ".join(e["kernel"] for e in fm["roofline_secondary"])
    assert "window_attn_kernel (HAB" in kinds and "window_attn_kernel (OCAB" in kinds and "attn_f32_kernel<32>" in kinds
    assert "part branch" in kinds and abs(fm["roofline"]["flops_per_launch"] - 4.0 * (8 * 1449) ** 2 * 1024) < 1.0


def test_bench_on_a_heavy_tailed_checkpoint_reports_the_rung():
    """`--weights` + `precision_rung`: on the sigma 1 / 0.5 dose 71 of 72 blocks escalate, the timed (escalated) model passes the
    dose fixture's check, the extra single-operand forward does not, and both step times are in the line."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--views", "8", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--weights", "trained_like(qk=1,norm=0.5)"], capture_output=True, text=True, cwd=ROOT,
                         timeout=600, env=dict(os.environ, IGGT_BENCH_WORSTCASE="0", IGGT_BENCH_BF16_LEG="0"))
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    rung = d["precision_rung"]
    assert rung["escalated"] == 71 and 0 < rung["ill_conditioned_by_own_figures"] < 71
    assert d["output_check"]["ok"] and d["output_check"]["fixture"].endswith("full_s8_518_tlD.pt")
    assert rung["single_fp16_output_check"]["max_l2"] > d["output_check"]["max_l2"]
    assert rung["ms_per_step"] > rung["ms_per_step_single_fp16_operands"] > 0


def test_bench_two_rank_control_flow():
    env = dict(os.environ, IGGT_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--views", "8", "--steps", "1",
           "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["graphs"] is True and "graphs_note" not in d
    assert d["config"]["parallelism"] == "view-shard x2"
    assert d["output_check"]["ok"] and d["output_check"]["max_l2_all_ranks"] < 1e-3
    assert "cpu_baseline" not in d            # rank 0 at N = 1 only


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset), as the driver types the N = 1 case: the script
    re-executes itself under torch.distributed.run and rank 0's JSON line comes back on stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["IGGT_BENCH_SINGLE_DEVICE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--views", "8", "--steps", "1",
                          "--warmup", "1"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "view-shard x2" and d["output_check"]["ok"]


def test_bench_self_launch_eight_ranks_one_device():
    """`python bench.py --gpus 8 --views 8` through the self-launch, all eight ranks on cuda:0 over gloo: the control flow of
    BASELINE.json configs[3] (one view per rank here, hipGraph segments, max-over-ranks timing) with every rank's outputs
    checked against its slice of the reference fixture and the worst rank reported."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["IGGT_BENCH_SINGLE_DEVICE"] = "1"
    env["IGGT_BENCH_BF16_LEG"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--views", "8", "--steps", "1",
                          "--warmup", "1"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 8 and d["graphs"] is True and d["config"]["parallelism"] == "view-shard x8"
    assert d["output_check"]["ok"] and d["output_check"]["max_l2_all_ranks"] < 1e-3
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_gpus8_views8_single_device.json"), "w") as f:
        json.dump(d, f)


def test_bench_emulated_rank_line():
    """`--emulate-world W` (developer mode): one middle rank of a W-GPU run on this GPU as a bench-shaped record -- hipGraph
    segments on, per-rank attention shape in the roofline entry, no output check (the other ranks' keys are copies)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--views", "8", "--emulate-world", "4", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, cwd=ROOT, timeout=900,
                         env=dict(os.environ, IGGT_BENCH_BF16_LEG="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    e = d["emulated_rank"]
    assert d["n_gpus"] == 1 and d["graphs"] is True and e["world"] == 4 and e["rank"] == 2 and e["views_of_this_rank"] == 2
    assert "output_check" not in d and "cpu_baseline" not in d
    assert abs(d["roofline"]["flops_per_launch"] - 4.0 * (2 * 1374) * (8 * 1374) * 1024) < 1.0
    assert abs(e["job_views_per_s_if_transport_were_free"] - 4 * d["value"]) < 1e-6 * d["value"] * 4 + 1e-9
