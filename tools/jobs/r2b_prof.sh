#!/bin/bash
# round-2 profiles of the default bench workload (64 x n=2000 x d=10, m=10000): kernel stats, FETCH/WRITE PMC, SQ PMC
export TMPDIR=/tmp
R=/root/repo
cd /tmp
rm -rf $R/gpurun_out/r2bp_stats $R/gpurun_out/r2bp_sq $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2bp_stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shard-sweep > $R/gpurun_out/r2bp_bench_under_prof.json 2> $R/gpurun_out/r2bp_stats.err
cd $R
python tools/prof_summary.py $(find gpurun_out/r2bp_stats -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shard-sweep" > gpurun_out/r02b_kernel_stats.txt 2>&1
head -30 gpurun_out/r02b_kernel_stats.txt | cut -c1-190
bash tools/pmc_fetch.sh gpurun_out/r02b_pmc_fetch_write_kb.txt
cd /tmp
PMC_M=10000 timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/r2bp_sq -- python $R/tools/pmc_step.py > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $(find gpurun_out/r2bp_sq -name "*.db") > gpurun_out/r02b_pmc_sq.txt
head -14 gpurun_out/r02b_pmc_sq.txt | cut -c1-230
