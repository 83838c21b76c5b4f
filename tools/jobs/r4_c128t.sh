#!/bin/bash
# round 4: A/B of the in-tree library against $REF on chain-bound and full launches + per-task traces of the chain-bound ones
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/c128ab_${TAG:-head}; mkdir -p $O
REF=$REF TAG=$TAG bash tools/jobs/r4_c128ab.sh > /dev/null 2>&1
for CFG in 1:2000:10 8:2000:10; do
  rm -f /tmp/mc.trace
  MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=$CFG REPS=1 timeout 300 python tools/mchol_check.py > /dev/null 2>&1
  python tools/mchol_trace.py /tmp/mc.trace -2 0 > $O/mchol_trace_${CFG//:/_}.txt 2>&1
done
