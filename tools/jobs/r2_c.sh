#!/bin/bash
# sweep + per-grid kernel times
export TMPDIR=/tmp
python tools/shard_sweep.py > gpurun_out/r2c_sweep.txt 2>&1
cut -c1-200 gpurun_out/r2c_sweep.txt
cd /tmp
rm -rf /root/repo/gpurun_out/r2c_prof
BS=${PBS:-8} REPS=5 timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2c_prof -- python /root/repo/tools/shard_sweep.py > /root/repo/gpurun_out/r2c_prof.log 2>&1
cd /root/repo
python tools/prof_summary.py $(find gpurun_out/r2c_prof -name "*.db" | head -1) > gpurun_out/r2c_prof_summary.txt 2>&1
head -24 gpurun_out/r2c_prof_summary.txt | cut -c1-200
