#!/bin/bash
# round-4 GPU pass K: 192-row variant of the two-workgroups-per-CU GEMM at the per-rank shapes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_f16_gpu.py -x -q -k "gemm" > gpurun_out/k_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/k_tests.log
timeout 600 python probes/gemm_rank_ab.py > gpurun_out/k_gemm_rank_ab.txt 2>&1
tail -4 gpurun_out/k_tests.log; cat gpurun_out/k_gemm_rank_ab.txt
