#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
for CFG in ${CFGS:-8:2000:10}; do
  rm -f /tmp/mc.trace
  MOGP_CHOL=mchol MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=$CFG REPS=1 timeout 300 python tools/mchol_check.py 2>&1 | tail -1 | cut -c1-120
  python tools/mchol_trace.py /tmp/mc.trace -2 0 > gpurun_out/r3c_trace_${CFG//:/_}.txt 2>&1
  cat gpurun_out/r3c_trace_${CFG//:/_}.txt | cut -c1-260
done
if [ -n "$FULL" ]; then MOGP_MC_SPIN=400000 CONFIGS=${FULL} timeout 600 python tools/mchol_check.py 2>&1 | tail -14 | cut -c1-200; fi
