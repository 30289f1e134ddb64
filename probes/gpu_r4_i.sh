#!/bin/bash
# round-4 GPU pass I: PMC counters on the fc1 GEMM (table vs polynomial GELU) and on the estimated-shift attention launch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
( IGGT_GELU_LUT=1 bash probes/pmc_kernel.sh $R/gpurun_out/pmc_fc1_lut.txt gemm_bf16_t256pp $R/probes/gemm_fc1_only.py ) > gpurun_out/i1.log 2>&1
( IGGT_GELU_LUT=0 bash probes/pmc_kernel.sh $R/gpurun_out/pmc_fc1_poly.txt gemm_bf16_t256pp $R/probes/gemm_fc1_only.py ) > gpurun_out/i2.log 2>&1
( bash probes/pmc_kernel.sh $R/gpurun_out/pmc_attn_est_sinks.txt attn $R/probes/attn_est_regime.py sinks ) > gpurun_out/i3.log 2>&1
( bash probes/pmc_kernel.sh $R/gpurun_out/pmc_attn_static.txt flash_attn $R/probes/attn_static_only.py static f16 3 ) > gpurun_out/i4.log 2>&1
{ echo "# rocprofv3 PMC passes, round 4: fc1 + GELU GEMM (M = 43 968, N = 4 096, K = 1 024, fp16) with the LDS-table GELU"; cat gpurun_out/pmc_fc1_lut.txt; echo; echo "# ... and with the polynomial erfc GELU (IGGT_GELU_LUT=0)"; cat gpurun_out/pmc_fc1_poly.txt; } > gpurun_out/r04_gemm_pmc.txt
{ echo "# rocprofv3 PMC passes, round 4: estimated-shift launch (forced) on the 'sinks' regime, fp16, N = 43 968 (probes/attn_est_regime.py)"; cat gpurun_out/pmc_attn_est_sinks.txt; echo; echo "# ... and the norm-bound static kernel on LayerNorm-of-noise operands (probes/attn_static_only.py)"; cat gpurun_out/pmc_attn_static.txt; } > gpurun_out/r04_attn_est_pmc.txt
echo done
