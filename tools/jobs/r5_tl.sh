#!/bin/bash
# round 5: command timeline (kernels + copies) of one evaluation: C2, the 8-emulator shard, the full batch
export TMPDIR=/tmp
cd /tmp
O=/root/repo/gpurun_out/r5tl; rm -rf $O; mkdir -p $O
for cfg in "1 2000 10" "8 2000 10" "64 2000 10"; do
  set -- $cfg
  rm -rf /tmp/tl_$1
  B=$1 N=$2 D=$3 REPS=6 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$1 -o tl -- python /root/repo/tools/mchol_time.py > /dev/null 2>&1
  echo "== B=$1 n=$2" | tee -a $O/timeline.txt
  python /root/repo/tools/timeline.py /tmp/tl_$1 2>&1 | tee -a $O/timeline.txt; find /tmp/tl_$1 -type f | head -5
done
