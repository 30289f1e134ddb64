#!/bin/bash
# round-4 GPU pass L: whole GPU suite on 3 xdist workers (early check of the kernel changes since pass H), emulated rank of 8
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -q -m gpu -n 3 --timeout 400 --durations=30 ) > gpurun_out/l1_suite.log 2>&1; echo "suite rc=$? $(tail -4 gpurun_out/l1_suite.log | tr '\n' ' ' | cut -c1-300)"
( time timeout 300 python bench.py --emulate-world 8 --steps 20 --warmup 5 ) > gpurun_out/l2_emu8.log 2>&1; echo "emu8 rc=$?"; grep '^{' gpurun_out/l2_emu8.log | tail -1 | cut -c1-400
