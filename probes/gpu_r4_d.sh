#!/bin/bash
# round-4 GPU pass D: est v2 tests, robustness table, per-kernel trace of the est launch, graphed many-rank shard runs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; ( time timeout "$TMO" "$@" ) > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -3 gpurun_out/$name.log | tr '\n' ' ')"; }
TMO=400 run d1_attn_est python -m pytest tests/test_attn_est_gpu.py -q
TMO=300 run d2_robust python probes/attn_static_robustness.py
cp gpurun_out/d2_robust.log gpurun_out/r04_attn_static_robustness.txt
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_est -o est -- python "$OLDPWD/probes/attn_est_profile.py" ) > gpurun_out/d3_prof.log 2>&1
find /tmp/prof_est -name "*kernel_stats*" | head -1 | xargs -I{} cp {} gpurun_out/r04_attn_est_kernel_stats.csv
TMO=300 run d4_trained python -m pytest tests/test_trained_like_gpu.py -q
TMO=600 run d5_shard python -m pytest tests/test_shard_gpu.py -q -k "many_rank and (8-True-True or 4-True-False)"
echo done
