# Round 6, GPU batch D: after removing the fused compensation launch; part branch in one pass, fused tail, two-pass convolutions.
mkdir -p gpurun_out
python -m pytest tests/test_conv_gpu.py -k "dpt_tail" tests/test_kernels_f16_gpu.py -k "gelu or colmean or dpt_tail" -q -m gpu -p no:cacheprovider 2>&1 | tail -3
python -m pytest "tests/test_headline_gpu.py::test_full_model_8_views_532_matches_reference" "tests/test_headline_gpu.py::test_full_model_32_views_532_matches_reference" "tests/test_headline_gpu.py::test_forward_2_views_1036_matches_reference" tests/test_e2e_gpu.py tests/test_checkpoint_gpu.py "tests/test_trained_like_gpu.py::test_outlier_channels_pass_on_single_operands_and_are_not_escalated" -q -m gpu -p no:cacheprovider > gpurun_out/r06_d_pytest.log 2>&1
tail -30 gpurun_out/r06_d_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_n1_d.json 2> gpurun_out/r06_bench_n1_d.err
IGGT_PART_CONV_PREC=3 IGGT_BENCH_WORSTCASE=0 IGGT_BENCH_BF16_LEG=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_n1_d_partprec3.json 2> gpurun_out/r06_bench_n1_d_partprec3.err
IGGT_BENCH_BF16_LEG=0 python bench.py --emulate-world 8 --steps 20 --warmup 5 > gpurun_out/r06_bench_emu8_d.json 2> gpurun_out/r06_bench_emu8_d.err
python - <<'PY'
import json
def load(f):
    try:
        return json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-800:])
d = load("gpurun_out/r06_bench_n1_d.json")
if d:
    print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["output_check"]["max_l2"])
    print("worst", {k: (round(v["ms_per_launch"], 3), v["mode"], v["rows_handed_over"]) for k, v in d["roofline_worstcase"]["per_regime"].items()})
    for e in d["roofline_secondary"]:
        print("  ", e["kernel"][:50], round(e["ms_per_forward"], 2), round(e["frac"], 4), {k: round(v["tflops"]) for k, v in e.get("per_shape", {}).items()})
for f in ("gpurun_out/r06_bench_n1_d.json", "gpurun_out/r06_bench_n1_d_partprec3.json"):
    d = load(f)
    if not d:
        continue
    fm = d.get("full_model", {})
    print(f, "full_model", {k: fm.get(k) for k in ("value", "ms_per_step", "part_branch_ms_per_forward", "peak_memory_gib", "error")})
    print("   check", {k: round(v["l2"], 6) for k, v in fm.get("output_check", {}).get("errors", {}).items()})
    for e in fm.get("roofline_secondary", []):
        print("     ", e["kernel"][:60], round(e["ms_per_forward"], 3), "ms", round(e["achieved"], 1), e["unit"], round(e["frac"], 4), e.get("mfma_passes_per_product"))
d = load("gpurun_out/r06_bench_emu8_d.json")
if d:
    print("emu8", d["ms_per_step"], d["roofline"]["ms_per_launch"], [(e["kernel"][:20], round(e["ms_per_forward"], 2), {k: round(v["tflops"]) for k, v in e.get("per_shape", {}).items()}) for e in d["roofline_secondary"]])
PY
bash probes/profile_cmd.sh gpurun_out/r06_full532_s32_kernel_stats.txt $PWD/probes/run_full.py 32 532 532 3 > /dev/null 2>&1
head -45 gpurun_out/r06_full532_s32_kernel_stats.txt | cut -c1-170
