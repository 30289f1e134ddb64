# Round 6, GPU batch G: part-head MLPs on the 16-bit GEMM path, HAB residual sum in fewer passes.
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -k "window_attn" "tests/test_headline_gpu.py::test_full_model_8_views_532_matches_reference" "tests/test_headline_gpu.py::test_full_model_32_views_532_matches_reference" "tests/test_headline_gpu.py::test_forward_2_views_1036_matches_reference" tests/test_e2e_gpu.py tests/test_real_images_gpu.py tests/test_graphs_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/r06_g_pytest.log 2>&1
tail -25 gpurun_out/r06_g_pytest.log
IGGT_BENCH_WORSTCASE=0 IGGT_BENCH_BF16_LEG=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06_bench_n1_g.json 2> gpurun_out/r06_bench_n1_g.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r06_bench_n1_g.json") if l.startswith("{")][-1])
    fm = d.get("full_model", {})
    print("headline", d["value"], d["ms_per_step"], "full_model", {k: fm.get(k) for k in ("value", "ms_per_step", "part_branch_ms_per_forward", "peak_memory_gib", "error")})
    print("   check", {k: round(v["l2"], 6) for k, v in fm.get("output_check", {}).get("errors", {}).items()})
    for e in fm.get("roofline_secondary", []):
        print("     ", e["kernel"][:60], round(e["ms_per_forward"], 3), "ms", round(e["achieved"], 1), e["unit"], round(e["frac"], 4), e.get("mfma_passes_per_product"))
except Exception as e:
    print("bench FAILED", e, open("gpurun_out/r06_bench_n1_g.err").read()[-1500:])
PY
python probes/part_branch_table.py > gpurun_out/r06_part_branch_table.txt 2>&1; tail -45 gpurun_out/r06_part_branch_table.txt
