#!/bin/bash
export TMPDIR=/tmp
cd /tmp
rm -rf /root/repo/gpurun_out/r2e_prof
env $CFG BS=${PBS:-8} REPS=3 timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r2e_prof -- python /root/repo/tools/shard_sweep.py > /root/repo/gpurun_out/r2e_prof.log 2>&1
cd /root/repo
python tools/timeline.py $(find gpurun_out/r2e_prof -name "*.db" | head -1) 3 120 > gpurun_out/r2e_timeline.txt 2>&1
cat gpurun_out/r2e_timeline.txt | cut -c1-150
