#!/bin/bash
# round 2, run B: new diagonal-block kernel -- parity tests, shard sweep, kernel trace
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2b_tests.txt
cat gpurun_out/r2b_tests.txt
python tools/shard_sweep.py > gpurun_out/r2b_sweep.txt 2>&1
cat gpurun_out/r2b_sweep.txt
cd /tmp
rm -rf /root/repo/gpurun_out/r2b_prof
BS=8,64 REPS=5 timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2b_prof -- python /root/repo/tools/shard_sweep.py > /root/repo/gpurun_out/r2b_prof.log 2>&1
cd /root/repo
python tools/prof_summary.py $(find gpurun_out/r2b_prof -name "*.db" | head -1) B=8,64 > gpurun_out/r2b_prof_summary.txt 2>&1
head -24 gpurun_out/r2b_prof_summary.txt
