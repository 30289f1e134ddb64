#!/bin/bash
# round-4 GPU pass H (final validation): whole GPU suite, smoke, bench with kernel trace, robustness table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; ( time timeout "$TMO" "$@" ) > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -3 gpurun_out/$name.log | tr '\n' ' ' | cut -c1-300)"; }
TMO=1500 run h1_suite python -m pytest tests -q -m gpu --timeout 400
TMO=200 run h2_smoke python -c "import __graft_entry__ as g; g.smoke()"
TMO=400 run h3_bench python bench.py --steps 20 --warmup 5
grep '^{' gpurun_out/h3_bench.log | tail -1 > gpurun_out/r04_bench_n1.json
TMO=300 run h4_robust python probes/attn_static_robustness.py
TMO=400 run h5_prof bash probes/profile_bench.sh $R/gpurun_out/r04_bench_n1_kernel_stats.txt
echo done
