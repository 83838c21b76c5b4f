#!/bin/bash
# generic A/B of library switches: results against the first configuration + mchol kernel time per configuration and shape
# usage: CFGS="a=1;b=2" SHAPES="64:2000:10 8:2000:10" TAG=x [TRACE=1] bash tools/jobs/r4_ab.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/ab_${TAG:-head}; mkdir -p $O
IFS=';' read -ra CF <<< "${CFGS}"
{
echo "== results vs the first configuration (64 x n=2000; fit, fit+grad)"
WHAT=fit,grad REPS=${REPS:-10} timeout 900 python tools/ab.py "${CF[@]}"
for shp in ${SHAPES:-64:2000:10}; do
  IFS=':' read -r b n d <<< "$shp"
  for cfg in "${CF[@]}"; do
    env $cfg B=$b N=$n D=$d REPS=${REPS:-10} timeout 300 python tools/mchol_time.py
  done
done
if [ -n "$TRACE" ]; then
  for cfg in "${CF[@]}"; do
    rm -f /tmp/mc.trace
    env $cfg MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=${TRACE_SHAPE:-64:2000:10} timeout 300 python tools/mchol_check.py > /dev/null 2>&1
    echo "== trace $cfg"
    python tools/mchol_trace.py /tmp/mc.trace -2 0 2>&1 | grep "^#"
  done
fi
} 2>&1 | grep -v "^$" | tee $O/ab.txt
