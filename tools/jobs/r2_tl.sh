#!/bin/bash
# kernel timelines of one fit (64 x n=2000) for the configurations in CFGS (";"-separated env strings)
export TMPDIR=/tmp
R=/root/repo
IFS=';' read -ra CF <<< "${CFGS:-MOGP_X=0}"
i=0
for c in "${CF[@]}"; do
  cd /tmp
  rm -rf $R/gpurun_out/tl_$i
  env $c timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$i -- python $R/tools/${SCRIPT:-fit_only.py} > $R/gpurun_out/tl_$i.log 2>&1
  cd $R
  echo "== $c: $(grep 'ms per' gpurun_out/tl_$i.log)"
  python tools/timeline.py $(find gpurun_out/tl_$i -name "*.db" | head -1) 12 400 > gpurun_out/tl_$i.txt 2>&1
  tail -1 gpurun_out/tl_$i.txt
  i=$((i+1))
done
