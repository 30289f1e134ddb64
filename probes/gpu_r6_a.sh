# Round 6, GPU batch A (run on the GPU box from the repository root): the new tests, the default bench line with its full_model
# leg, the kernel table of the whole model incl. part_feat at 32 x 532^2.
mkdir -p gpurun_out
python -m pytest tests/test_kernels_f16_gpu.py tests/test_checkpoint_gpu.py tests/test_attn_est_gpu.py "tests/test_headline_gpu.py::test_full_model_8_views_532_matches_reference" "tests/test_headline_gpu.py::test_full_model_32_views_532_matches_reference" tests/test_e2e_gpu.py "tests/test_bench_gpu.py::test_bench_single_gpu_line" -q -m gpu -p no:cacheprovider > gpurun_out/r06_a_pytest.log 2>&1
tail -25 gpurun_out/r06_a_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_n1_a.json 2> gpurun_out/r06_bench_n1_a.err
tail -c 1500 gpurun_out/r06_bench_n1_a.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_bench_n1_a.json") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["output_check"]["max_l2"])
fm = d.get("full_model", {})
print("full_model", {k: fm.get(k) for k in ("value", "ms_per_step", "part_branch_ms_per_forward", "peak_memory_gib", "error")})
print("full_model check", fm.get("output_check", {}).get("errors"))
for e in fm.get("roofline_secondary", []):
    print("  ", e["kernel"][:60], round(e["ms_per_forward"], 3), "ms", round(e["achieved"], 1), e["unit"], round(e["frac"], 4))
print("worst", {k: (round(v["ms_per_launch"], 3), v["mode"], v["rows_handed_over"]) for k, v in d["roofline_worstcase"]["per_regime"].items()})
for e in d["roofline_secondary"]:
    print("  ", e["kernel"][:50], round(e["ms_per_forward"], 2), round(e["frac"], 4), {k: round(v["tflops"]) for k, v in e.get("per_shape", {}).items()})
print("cpu", d.get("cpu_baseline"))
PY
bash probes/profile_cmd.sh gpurun_out/r06_full532_s32_kernel_stats.txt probes/run_full.py 32 532 532 3 > /dev/null 2>&1
head -40 gpurun_out/r06_full532_s32_kernel_stats.txt | cut -c1-160
