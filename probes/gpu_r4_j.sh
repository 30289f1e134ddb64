#!/bin/bash
# round-4 GPU pass J: estimated-shift kernel with rows dealt to the lanes in shift order + per-wave choice of the tile loop
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attn_est_gpu.py -x -q > gpurun_out/j_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/j_tests.log
timeout 600 python probes/attn_static_robustness.py > gpurun_out/j_robust.txt 2>&1
tail -5 gpurun_out/j_tests.log; cat gpurun_out/j_robust.txt | tail -40
