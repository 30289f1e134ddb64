#!/bin/bash
# round-4 GPU pass M (final validation): whole GPU suite in the driver's order, smoke, bench + its kernel trace, est-launch traces
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; ( time timeout "$TMO" "$@" ) > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -3 gpurun_out/$name.log | tr '\n' ' ' | cut -c1-300)"; }
TMO=1100 run m1_suite python -m pytest tests -x -q -m gpu --timeout 400
TMO=200 run m2_smoke python -c "import __graft_entry__ as g; g.smoke()"
TMO=400 run m3_bench python bench.py --steps 20 --warmup 5
grep '^{' gpurun_out/m3_bench.log | tail -1 > gpurun_out/r04_bench_n1.json
TMO=300 run m4_prof bash probes/profile_bench.sh $R/gpurun_out/r04_bench_n1_kernel_stats.txt
for k in registers affine sinks; do
  TMO=120 run m5_prof_$k bash probes/profile_cmd.sh $R/gpurun_out/r04_attn_est_trace_$k.txt $R/probes/attn_est_regime.py $k
  grep "rows handed over" /tmp/prof_cmd.log >> gpurun_out/r04_attn_est_trace_$k.txt
done
echo done
