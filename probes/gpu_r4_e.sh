#!/bin/bash
# round-4 GPU pass E: kernel trace of the est launch, robustness, full GPU suite, bench (N=1, emulated rank of 8, config 5 whole)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; ( time timeout "$TMO" "$@" ) > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -3 gpurun_out/$name.log | tr '\n' ' ' | cut -c1-300)"; }
TMO=200 run e1_prof bash probes/profile_cmd.sh $R/gpurun_out/r04_attn_est_kernel_stats.txt $R/probes/attn_est_profile.py
TMO=200 run e2_robust python probes/attn_static_robustness.py
cp gpurun_out/e2_robust.log gpurun_out/r04_attn_static_robustness.txt
TMO=1500 run e3_suite python -m pytest tests -q -m gpu --timeout 400
TMO=400 run e4_bench python bench.py --steps 10 --warmup 3
grep '^{' gpurun_out/e4_bench.log | tail -1 > gpurun_out/r04_bench_n1.json
TMO=300 run e5_emu8 python bench.py --emulate-world 8 --steps 10 --warmup 3
grep '^{' gpurun_out/e5_emu8.log | tail -1 > gpurun_out/r04_bench_emu8_config4.json
export IGGT_BENCH_WORSTCASE=0 IGGT_BENCH_BF16_LEG=0
TMO=500 run e6_config5 python bench.py --views 64 --size 1036 --steps 1 --warmup 0 --no-cpu-baseline
grep '^{' gpurun_out/e6_config5.log | tail -1 > gpurun_out/r04_config5_full_n1.json
echo done
