# Round 6, GPU batch F: profiles of the final code.
mkdir -p gpurun_out
python probes/part_branch_table.py > gpurun_out/r06_part_branch_table.txt 2>&1; tail -40 gpurun_out/r06_part_branch_table.txt
bash probes/profile_bench.sh gpurun_out/r06_bench_n1_kernel_stats.txt > /dev/null 2>&1; head -30 gpurun_out/r06_bench_n1_kernel_stats.txt | cut -c1-160
bash probes/profile_cmd.sh gpurun_out/r06_full532_s32_kernel_stats.txt $PWD/probes/run_full.py 32 532 532 3 > /dev/null 2>&1
bash probes/pmc_traffic.sh gpurun_out/r06_attn_traffic_pmc_n1.txt flash_attn probes/attn_static_only.py > /dev/null 2>&1; cat gpurun_out/r06_attn_traffic_pmc_n1.txt
bash probes/pmc_traffic.sh gpurun_out/r06_attn_traffic_pmc_rank8.txt attn probes/attn_rank_only.py config4 > /dev/null 2>&1; cat gpurun_out/r06_attn_traffic_pmc_rank8.txt
bash probes/pmc_traffic.sh gpurun_out/r06_attn_traffic_pmc_config5.txt attn probes/attn_rank_only.py config5 > /dev/null 2>&1; cat gpurun_out/r06_attn_traffic_pmc_config5.txt
bash probes/pmc_kernel.sh gpurun_out/r06_window_attn_pmc.txt window_attn probes/run_full.py 8 532 532 2 > /dev/null 2>&1; cat gpurun_out/r06_window_attn_pmc.txt
