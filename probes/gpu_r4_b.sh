#!/bin/bash
# round-4 GPU pass B: estimated-shift attention after the pairing / key-scan fixes; bisecting the graphed shard fault
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; ( time timeout "$TMO" "$@" ) > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -3 gpurun_out/$name.log | tr '\n' ' ')"; }
TMO=400 run b1_attn_est python -m pytest tests/test_attn_est_gpu.py -q
TMO=300 run b2_robust python probes/attn_static_robustness.py
cp gpurun_out/b2_robust.log gpurun_out/r04_attn_static_robustness.txt
TMO=300 run b3_two_rank python -m pytest tests/test_shard_gpu.py -q -k "two_rank"
TMO=900 run b4_debug python probes/shard_debug.py single:1 single:2 4:1:0:IGGT_ATTN_EST=0 4:1:0 4:0:1
echo done
