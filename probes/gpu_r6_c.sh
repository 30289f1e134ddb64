# Round 6, GPU batch C: fused compensation A/B (numerics at the tiny fixtures, launch times), kernel table of the whole model.
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
IGGT_COMP_FUSED=0 python -m pytest tests/test_e2e_gpu.py -k "forward_matches_reference" -q -m gpu -p no:cacheprovider > gpurun_out/r06_c_e2e_fused0.log 2>&1
cp gpurun_out/parity_report.json gpurun_out/r06_c_parity_fused0.json; rm -f gpurun_out/parity_report.json
python -m pytest tests/test_e2e_gpu.py -k "forward_matches_reference" -q -m gpu -p no:cacheprovider > gpurun_out/r06_c_e2e_fused1.log 2>&1
cp gpurun_out/parity_report.json gpurun_out/r06_c_parity_fused1.json
tail -3 gpurun_out/r06_c_e2e_fused0.log gpurun_out/r06_c_e2e_fused1.log
python -m pytest tests/test_kernels_f16_gpu.py -k "comp_bias or gelu_table" "tests/test_headline_gpu.py::test_full_model_32_views_532_matches_reference" -q -m gpu -p no:cacheprovider 2>&1 | tail -5
python probes/comp_time.py > gpurun_out/r06_comp_bias_ab.txt 2>&1; cat gpurun_out/r06_comp_bias_ab.txt
IGGT_BENCH_BF16_LEG=0 python bench.py --emulate-world 8 --steps 20 --warmup 5 > gpurun_out/r06_bench_emu8_c.json 2> gpurun_out/r06_bench_emu8_c.err
IGGT_COMP_FUSED=0 IGGT_BENCH_BF16_LEG=0 python bench.py --emulate-world 8 --steps 20 --warmup 5 > gpurun_out/r06_bench_emu8_c_fused0.json 2> gpurun_out/r06_bench_emu8_c_fused0.err
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench_emu8_c.json", "gpurun_out/r06_bench_emu8_c_fused0.json"):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, d["ms_per_step"], d["roofline"]["ms_per_launch"], [(e["kernel"][:20], round(e["ms_per_forward"], 2), {k: round(v["tflops"]) for k, v in e.get("per_shape", {}).items()}) for e in d["roofline_secondary"]])
    except Exception as e:
        print(f, "FAILED", e)
a, b = json.load(open("gpurun_out/r06_c_parity_fused0.json")), json.load(open("gpurun_out/r06_c_parity_fused1.json"))
for k in sorted(a):
    if k in b and "tiny" in k:
        print(k, {q: (round(a[k][q]["l2"], 7), round(b[k][q]["l2"], 7)) for q in ("tokens_23", "depth", "depth_conf", "world_points") if q in a[k]},
              "centred depth_conf", round(a[k]["depth_conf"]["l2_centered"], 6), round(b[k]["depth_conf"]["l2_centered"], 6))
PY
bash probes/profile_cmd.sh gpurun_out/r06_full532_s32_kernel_stats.txt $PWD/probes/run_full.py 32 532 532 3 > /dev/null 2>&1
head -75 gpurun_out/r06_full532_s32_kernel_stats.txt | cut -c1-170; tail -8 gpurun_out/r06_full532_s32_kernel_stats.txt
