#!/bin/bash
# round-4 GPU pass G: per-kernel trace of the est launch per regime, HBM-side traffic of the config-5 per-rank attention, test subset
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; ( time timeout "$TMO" "$@" ) > gpurun_out/$name.log 2>&1; echo "rc=$? $(tail -3 gpurun_out/$name.log | tr '\n' ' ' | cut -c1-300)"; }
for k in registers affine sinks; do
  TMO=120 run g1_prof_$k bash probes/profile_cmd.sh $R/gpurun_out/r04_attn_est_trace_$k.txt $R/probes/attn_est_regime.py $k
  grep "rows handed over" /tmp/prof_cmd.log >> gpurun_out/r04_attn_est_trace_$k.txt
done
TMO=400 run g2_pmc5 bash probes/pmc_traffic.sh $R/gpurun_out/r04_attn_traffic_pmc_config5.txt attn $R/probes/attn_rank_only.py config5 f16 2
TMO=600 run g3_tests python -m pytest tests/test_attn_est_gpu.py tests/test_e2e_gpu.py tests/test_graphs_gpu.py tests/test_trained_like_gpu.py -q --timeout 300
echo done
