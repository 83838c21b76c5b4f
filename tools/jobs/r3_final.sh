#!/bin/bash
# round 3, final profiles of HEAD in ONE job.  usage: TAG=<commit> bash tools/jobs/r3_final.sh
# Every file goes to gpurun_out/final_$TAG/ and is copied into profiles/r03_$TAG_* afterwards.
export TMPDIR=/tmp
R=/root/repo
TAG=${TAG:-head}
O=$R/gpurun_out/final_$TAG
rm -rf $O; mkdir -p $O
cd $R
# 1. the GPU suite
timeout 1700 python -m pytest tests -m gpu -x -q --durations=10 2>&1 | tail -22 > $O/gpu_tests.txt
tail -3 $O/gpu_tests.txt
# 2. the bench line the driver will also produce
python bench.py > $O/bench.json 2> $O/bench.err
# 3. kernel stats of the default bench workload
cd /tmp
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shard-sweep --no-other-configs"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shard-sweep --no-other-configs > $O/bench_under_prof.json 2> $O/stats.err
cd $R
python tools/prof_summary.py $(find $O/stats -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- $CMD" > $O/kernel_stats.txt 2>&1
head -24 $O/kernel_stats.txt | cut -c1-190
# 4. C4 / C5 kernel stats
cd /tmp
for C in C4 C5; do
  ONLY=$C timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$C -- python $R/tools/big_configs.py > $O/prof_$C.log 2>&1
  python $R/tools/prof_summary.py $(find $O/prof_$C -name "*.db" | head -1) "ONLY=$C rocprofv3 --kernel-trace --stats -- python tools/big_configs.py" > $O/${C}_kernel_stats.txt 2>&1
  grep "^\[C" $O/prof_$C.log >> $O/${C}_kernel_stats.txt
  head -12 $O/${C}_kernel_stats.txt | cut -c1-170
done
# 5. L2-miss traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters (own pass), default workload
cd $R
bash tools/pmc_fetch.sh $O/pmc_fetch_write_kb.txt
# 5b. the lock-step form of the predictive variance (MOGP_PV_SYNC): same two passes, and its time against the default in the same job
bash tools/pmc_fetch.sh $O/pmc_fetch_write_kb_pv_sync.txt MOGP_PV_SYNC=1000
WHAT=predict REPS=8 timeout 600 python tools/ab.py "" "MOGP_PV_SYNC=1000" "" "MOGP_PV_SYNC=1000" 2>&1 | tail -4 > $O/pv_sync_ab.txt
cat $O/pv_sync_ab.txt
cd /tmp
PMC_M=10000 timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/sq -- python $R/tools/pmc_step.py > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $(find $O/sq -name "*.db") > $O/pmc_sq.txt
head -12 $O/pmc_sq.txt | cut -c1-230
# 6. per-task traces of the one-launch Cholesky: shard (8 emulators), full batch, C5
for CFG in 8:2000:10 64:2000:10 1:16000:8; do
  rm -f /tmp/mc.trace
  MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=$CFG REPS=1 timeout 300 python tools/mchol_check.py > /dev/null 2>&1
  python tools/mchol_trace.py /tmp/mc.trace -2 0 > $O/mchol_trace_${CFG//:/_}.txt 2>&1
done
head -8 $O/mchol_trace_8_2000_10.txt | cut -c1-200
rm -rf $O/stats $O/prof_C4 $O/prof_C5 $O/sq     # (raw databases stay on the box)
ls -la $O
