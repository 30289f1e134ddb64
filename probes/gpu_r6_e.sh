# Round 6, GPU batch E: MFMA window attention; the whole GPU suite.
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -k "window_attn" -q -m gpu -p no:cacheprovider 2>&1 | tail -15
IGGT_BENCH_WORSTCASE=0 IGGT_BENCH_BF16_LEG=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06_bench_n1_e.json 2> gpurun_out/r06_bench_n1_e.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r06_bench_n1_e.json") if l.startswith("{")][-1])
    fm = d.get("full_model", {})
    print("headline", d["value"], d["ms_per_step"], "full_model", {k: fm.get(k) for k in ("value", "ms_per_step", "part_branch_ms_per_forward", "peak_memory_gib", "error")})
    print("   check", {k: round(v["l2"], 6) for k, v in fm.get("output_check", {}).get("errors", {}).items()})
    for e in fm.get("roofline_secondary", []):
        print("     ", e["kernel"][:60], round(e["ms_per_forward"], 3), "ms", round(e["achieved"], 1), e["unit"], round(e["frac"], 4), e.get("mfma_passes_per_product"))
except Exception as e:
    print("bench FAILED", e, open("gpurun_out/r06_bench_n1_e.err").read()[-1500:])
PY
rm -f gpurun_out/parity_report.json
python -m pytest tests -q -m gpu -p no:cacheprovider -x --durations=15 > gpurun_out/r06_e_suite.log 2>&1
tail -40 gpurun_out/r06_e_suite.log
