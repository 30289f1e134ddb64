#!/bin/bash
# round-4 GPU pass A: new attention machinery, GELU table, trained_like fixtures, many-rank shard tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; ( time timeout "$TMO" "$@" ) > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -3 gpurun_out/$name.log | tr '\n' ' ')"; }
TMO=400 run a1_attn_est python -m pytest tests/test_attn_est_gpu.py -q -x
TMO=400 run a2_kernels_f16 python -m pytest tests/test_kernels_f16_gpu.py -q -k "gelu or epilogues or static"
TMO=300 run a3_robust python probes/attn_static_robustness.py
cp gpurun_out/a3_robust.log gpurun_out/r04_attn_static_robustness.txt
TMO=200 run a4_gelu_ab python probes/gemm_gelu_ab.py 32 8
TMO=600 run a5_trained_like python -m pytest tests/test_trained_like_gpu.py -q
TMO=600 run a6_shard python -m pytest tests/test_shard_gpu.py -q -k "many_rank"
echo done
