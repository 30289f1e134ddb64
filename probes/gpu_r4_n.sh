#!/bin/bash
# round-4 GPU pass N: refresh of the est-launch kernel table and of the robustness table after the last kernel changes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 bash probes/profile_cmd.sh $R/gpurun_out/r04_attn_est_kernel_stats.txt $R/probes/attn_est_profile.py > gpurun_out/n1_prof.log 2>&1
cd $R
timeout 300 python probes/attn_static_robustness.py 2>&1 | grep -v amdgpu.ids > gpurun_out/n2_robust.txt
cat gpurun_out/n2_robust.txt | cut -c1-150; head -12 gpurun_out/r04_attn_est_kernel_stats.txt | cut -c1-170
