# Round 6, final validation on the GPU box (from the repository root): whole GPU suite, smoke, the default bench line, emulated rank,
# kernel tables, window-attention counters, config 5 on one GPU.  Everything it writes lands in gpurun_out/; the copies under profiles/ are named r06_*.
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > gpurun_out/r06_h_suite.log 2>&1
tail -25 gpurun_out/r06_h_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_h_smoke.log 2>&1; tail -7 gpurun_out/r06_h_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err
IGGT_BENCH_BF16_LEG=0 python bench.py --emulate-world 8 --steps 20 --warmup 5 > gpurun_out/r06_bench_emu8.json 2> gpurun_out/r06_bench_emu8.err
python - <<'PY'
import json
def load(f):
    try:
        return json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-800:])
d = load("gpurun_out/r06_bench_n1.json")
if d:
    print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"], d["output_check"]["max_l2"], "bf16", d.get("roofline_bf16", {}).get("frac"))
    print("worst", d["roofline_worstcase"]["frac"], {k: (round(v["ms_per_launch"], 3), v["mode"], v["rows_handed_over"]) for k, v in d["roofline_worstcase"]["per_regime"].items()})
    for e in d["roofline_secondary"]:
        print("  ", e["kernel"][:50], round(e["ms_per_forward"], 2), round(e["frac"], 4), {k: round(v["tflops"]) for k, v in e.get("per_shape", {}).items()})
    fm = d.get("full_model", {})
    print("full_model", {k: fm.get(k) for k in ("value", "ms_per_step", "part_branch_ms_per_forward", "peak_memory_gib", "error")}, fm.get("roofline", {}).get("frac"))
    print("   check", {k: round(v["l2"], 6) for k, v in fm.get("output_check", {}).get("errors", {}).items()})
    for e in fm.get("roofline_secondary", []):
        print("     ", e["kernel"][:60], round(e["ms_per_forward"], 3), "ms", round(e["achieved"], 1), e["unit"], round(e["frac"], 4), e.get("mfma_passes_per_product"))
    print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind")})
d = load("gpurun_out/r06_bench_emu8.json")
if d:
    print("emu8", d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"], [(e["kernel"][:20], round(e["ms_per_forward"], 2), round(e["frac"], 4), {k: round(v["tflops"]) for k, v in e.get("per_shape", {}).items()}) for e in d["roofline_secondary"]])
PY
bash probes/profile_bench.sh gpurun_out/r06_bench_n1_kernel_stats.txt > /dev/null 2>&1
bash probes/profile_cmd.sh gpurun_out/r06_full532_s32_kernel_stats.txt $PWD/probes/run_full.py 32 532 532 3 > /dev/null 2>&1; head -30 gpurun_out/r06_full532_s32_kernel_stats.txt | cut -c1-165
timeout 600 bash probes/pmc_kernel.sh gpurun_out/r06_window_attn_pmc.txt window_attn probes/window_attn_only.py 32 3 > /dev/null 2>&1; cat gpurun_out/r06_window_attn_pmc.txt | cut -c1-250
timeout 300 python bench.py --views 64 --size 1036 --steps 2 --warmup 1 --no-cpu-baseline --no-full-model > gpurun_out/r06_config5_full_n1.json 2> gpurun_out/r06_config5_full_n1.err; python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r06_config5_full_n1.json") if l.startswith("{")][-1])
    print("config5 one GPU", d["ms_per_step"], d["value"], d["peak_memory_gib"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"])
except Exception as e:
    print("config5 FAILED", e, open("gpurun_out/r06_config5_full_n1.err").read()[-600:])
PY
