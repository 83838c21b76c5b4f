#!/bin/bash
# round 6, first job: the adaptive-nugget knife-edge on the device (default schedule, forced one-launch, forced multi-launch) + a bench line of HEAD
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_a; rm -rf $O; mkdir -p $O; cd $R
for c in default mchol left la right; do
  if [ $c = default ]; then timeout 300 python tools/adaptive_edge.py; else MOGP_CHOL=$c timeout 300 python tools/adaptive_edge.py; fi
done > $O/adaptive_edge.txt 2>&1
cat $O/adaptive_edge.txt | cut -c1-400
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json
