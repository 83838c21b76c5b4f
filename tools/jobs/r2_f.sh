#!/bin/bash
export TMPDIR=/tmp
tools/chol128_probe0.bin 64 | head -1
for cfg in "MOGP_X=1" "MOGP_LA_SERIAL=1" "MOGP_CHOL=left"; do
  echo "== $cfg"
  env $cfg BS=8,64 python tools/shard_sweep.py 2>&1 | cut -c1-140
done
cd /tmp
rm -rf /root/repo/gpurun_out/r2f_prof
MOGP_LA_SERIAL=1 BS=64 REPS=3 timeout 300 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r2f_prof -- python /root/repo/tools/shard_sweep.py > /root/repo/gpurun_out/r2f_prof.log 2>&1
cd /root/repo
python tools/timeline.py $(find gpurun_out/r2f_prof -name "*.db" | head -1) 3 70 > gpurun_out/r2f_timeline.txt 2>&1
cat gpurun_out/r2f_timeline.txt | cut -c1-140
