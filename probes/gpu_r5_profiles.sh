# Round-5 profile set (run on the GPU box from the repository root): kernel table of the bench, counters of this round's attention forms.
mkdir -p gpurun_out
bash probes/profile_bench.sh gpurun_out/r05_bench_n1_kernel_stats.txt > /dev/null 2>&1
bash probes/pmc_kernel.sh gpurun_out/r05_attn_x3_pmc.txt flash_attn_x3 probes/attn_round5_only.py x3 2 > /dev/null 2>&1
bash probes/pmc_kernel.sh gpurun_out/r05_attn_est_pmc.txt attn probes/attn_round5_only.py est 3 > /dev/null 2>&1
bash probes/pmc_kernel.sh gpurun_out/r05_attn_est_rank_pmc.txt attn probes/attn_round5_only.py rank 3 > /dev/null 2>&1
head -12 gpurun_out/r05_bench_n1_kernel_stats.txt; head -30 gpurun_out/r05_attn_x3_pmc.txt
