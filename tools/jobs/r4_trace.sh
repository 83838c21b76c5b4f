#!/bin/bash
# per-task traces of the one-launch Cholesky at 64 x n=2000 for a list of configurations; usage: CFGS="a=1;b=2 c=3" TAG=x bash tools/jobs/r4_trace.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/trace_${TAG:-head}; mkdir -p $O
i=0
IFS=';' read -ra CF <<< "${CFGS:-MOGP_MC_PAIR=0;MOGP_MC_PAIR=1}"
for cfg in "${CF[@]}"; do
  rm -f /tmp/mc.trace
  env $cfg MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=${SHAPE:-64:2000:10} timeout 300 python tools/mchol_check.py > /dev/null 2>&1
  echo "== $cfg" | tee -a $O/trace.txt
  python tools/mchol_trace.py /tmp/mc.trace -2 0 2>&1 | grep "^#" | tee -a $O/trace.txt
  i=$((i+1))
done
