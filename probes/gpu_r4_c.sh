#!/bin/bash
# round-4 GPU pass C: estimated-shift attention v2 (no atomics, per-row shifts, second chance); bisecting the graphed shard fault
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; ( time timeout "$TMO" "$@" ) > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -3 gpurun_out/$name.log | tr '\n' ' ')"; }
TMO=400 run c1_attn_est python -m pytest tests/test_attn_est_gpu.py -q
TMO=300 run c2_robust python probes/attn_static_robustness.py
cp gpurun_out/c2_robust.log gpurun_out/r04_attn_static_robustness.txt
T="tests/test_shard_gpu.py::test_two_rank_sharded_forward_matches_reference[tiny_s2_56_stress-1-True-False]"
for m in 63 0 62 61 59 55 47 15; do
  export IGGT_EST_DEBUG=$m
  TMO=200 run c3_mask$m python -m pytest "$T" -q -x
done
unset IGGT_EST_DEBUG
echo done
