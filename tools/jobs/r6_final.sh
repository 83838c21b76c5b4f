#!/bin/bash
# round 6: the whole profile set of HEAD in ONE job.  usage: TAG=<commit> bash tools/jobs/r6_final.sh
# Everything goes to gpurun_out/final_$TAG/; tools/collect_profiles.py <commit> files it under profiles/r06_<commit> (ROUND=r06 python tools/collect_profiles.py <commit>)_* -- and refuses when the
# library that ran here was not built from that commit (build stamp in $O/build_commit.txt).
export TMPDIR=/tmp
R=/root/repo
TAG=${TAG:-head}
O=$R/gpurun_out/final_$TAG
rm -rf $O; mkdir -p $O
cd $R
cp mogp_emulator_amd/libmogp_hip.build $O/build_commit.txt; cat $O/build_commit.txt        # written by the Makefile next to the library
# 1. the bench line the driver will also produce (first: on the fresh box, as the driver runs it)
python bench.py > $O/bench.json 2> $O/bench.err
# 2. the GPU suite + smoke
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 2>&1 | tail -22 > $O/gpu_tests.txt
tail -3 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/gpu_tests.txt
# 3. kernel stats of the default bench workload: ONLY the timed steps (+ the single-stream pass), no extras
cd /tmp
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shard-sweep --no-other-configs --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shard-sweep --no-other-configs --no-extras > $O/bench_under_prof.json 2> $O/stats.err
cd $R
python tools/prof_summary.py $(find $O/stats -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- $CMD" > $O/kernel_stats.txt 2>&1
head -16 $O/kernel_stats.txt | cut -c1-190
# 4. kernel stats of C2 (one n=2000 emulator), the 8-emulator shard, C4, C5
cd /tmp
for C in C2 S8 C4 C5; do
  ONLY=$C timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$C -- python $R/tools/big_configs.py > $O/prof_$C.log 2>&1
  python $R/tools/prof_summary.py $(find $O/prof_$C -name "*.db" | head -1) "ONLY=$C rocprofv3 --kernel-trace --stats -- python tools/big_configs.py" > $O/${C}_kernel_stats.txt 2>&1
  grep "^\[$C\|^     " $O/prof_$C.log >> $O/${C}_kernel_stats.txt
  head -9 $O/${C}_kernel_stats.txt | cut -c1-170
done
# 5. L2-miss traffic (FETCH_SIZE / WRITE_SIZE, separate passes): default workload, the 8-emulator shard, C2
cd $R
bash tools/pmc_fetch.sh $O/pmc_fetch_write_kb.txt
bash tools/pmc_fetch.sh $O/pmc_fetch_write_kb_S8.txt PMC_B=8
bash tools/pmc_fetch.sh $O/pmc_fetch_write_kb_C2.txt PMC_B=1
# 6. SQ counters (own pass each): default workload, shard, C2;  L2 hit / miss of the default workload
cd /tmp
for CFG in 64 8 1; do
  PMC_B=$CFG PMC_M=10000 timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/sq_$CFG -- python $R/tools/pmc_step.py > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find $O/sq_$CFG -name "*.db") > $O/pmc_sq_B$CFG.txt
done
PMC_M=10000 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/sqv -- python $R/tools/pmc_step.py > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find $O/sqv -name "*.db") > $O/pmc_sq_valu_B64.txt 2>&1
PMC_M=10000 timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc -- python $R/tools/pmc_step.py > $O/tcc.log 2>&1
python $R/tools/pmc_summary.py $(find $O/tcc -name "*.db") > $O/pmc_tcc.txt 2>&1
cd $R
head -8 $O/pmc_sq_B64.txt | cut -c1-230; head -6 $O/pmc_tcc.txt | cut -c1-160
# 7. the traffic-free A/B of the one-launch Cholesky and its per-task traces (shard, full batch, C5)
{ for cfg in "MOGP_MC_NOTRAFFIC=0" "MOGP_MC_NOTRAFFIC=1"; do env $cfg timeout 300 python tools/mchol_time.py; done; } > $O/mchol_notraffic.txt 2>&1
timeout 120 tools/gemm_loop_probe.bin 64 40 > $O/gemm_loop_probe.txt 2>&1
for CFG in 1:2000:10 8:2000:10 64:2000:10 1:16000:8; do
  rm -f /tmp/mc.trace
  MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=$CFG REPS=1 timeout 300 python tools/mchol_check.py > /dev/null 2>&1
  python tools/mchol_trace.py /tmp/mc.trace -2 0 > $O/mchol_trace_${CFG//:/_}.txt 2>&1
done
rm -rf $O/sqv $O/stats $O/prof_C2 $O/prof_S8 $O/prof_C4 $O/prof_C5 $O/sq_64 $O/sq_8 $O/sq_1 $O/tcc     # (raw databases stay on the box)
ls $O
