#!/bin/bash
# round 2, run A: baseline shard sweep + kernel trace for B=8
export TMPDIR=/tmp
python tools/shard_sweep.py > gpurun_out/r2a_sweep.txt 2>&1
SWEEP=c4 python tools/shard_sweep.py >> gpurun_out/r2a_sweep.txt 2>&1
cd /tmp
rm -rf /root/repo/gpurun_out/r2a_prof8
BS=8 REPS=5 timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2a_prof8 -- python /root/repo/tools/shard_sweep.py > /root/repo/gpurun_out/r2a_prof8.log 2>&1
cd /root/repo
python tools/prof_summary.py $(find gpurun_out/r2a_prof8 -name "*.db" | head -1) B=8 > gpurun_out/r2a_prof8_summary.txt 2>&1
cat gpurun_out/r2a_sweep.txt
head -30 gpurun_out/r2a_prof8_summary.txt
