#!/bin/bash
# round-4 GPU pass P: the 8-bit shift-difference threshold of the estimated-shift kernel -- its tests, the bench line, the robustness table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
( timeout 400 python -m pytest tests/test_attn_est_gpu.py tests/test_trained_like_gpu.py tests/test_graphs_gpu.py -x -q --timeout 300 ) > gpurun_out/p1_tests.log 2>&1; echo "tests rc=$? $(tail -2 gpurun_out/p1_tests.log | tr '\n' ' ')"
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/p2_bench.log 2>&1; echo "bench rc=$?"
grep '^{' gpurun_out/p2_bench.log | tail -1 > gpurun_out/r04_bench_n1.json
timeout 200 python probes/attn_static_robustness.py 2>&1 | grep -v amdgpu.ids > gpurun_out/p3_robust.txt
cut -c1-150 gpurun_out/p3_robust.txt
