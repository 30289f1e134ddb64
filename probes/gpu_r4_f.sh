#!/bin/bash
# round-4 GPU pass F: A/B of the estimated-shift instantiation, range-folding test, HDBSCAN parity on model features
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; ( time timeout "$TMO" "$@" ) > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -3 gpurun_out/$name.log | tr '\n' ' ' | cut -c1-300)"; }
TMO=400 run f1_est_ab python probes/attn_est_ab.py
TMO=300 run f2_precision python -m pytest tests/test_precision_gpu.py -q
TMO=400 run f3_hdbscan python -m pytest tests/test_post_gpu.py -q -k "model_features or timing_report"
echo done
