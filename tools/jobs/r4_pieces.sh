#!/bin/bash
# round 4: the k range of a GEMM task consumed in published 16-column pieces (MOGP_MC_PIECES) x number of chain-task row pairs (MOGP_MC_URG)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/pieces_${TAG:-head}; mkdir -p $O
{
for shp in ${SHAPES:-1:2000:10 8:2000:10 64:2000:10 2:5000:20}; do
  IFS=':' read -r b n d <<< "$shp"
  echo "== $shp"
  B=$b N=$n D=$d WHAT=fit REPS=${REPS:-15} timeout 900 python tools/ab.py "MOGP_MC_PIECES=0 MOGP_MC_URG=0" "MOGP_MC_PIECES=1 MOGP_MC_URG=0" "MOGP_MC_PIECES=1 MOGP_MC_URG=1" "MOGP_MC_PIECES=1 MOGP_MC_URG=2" "MOGP_MC_PIECES=0 MOGP_MC_URG=2" ""
done
} 2>&1 | grep -v "^$" | tee $O/ab.txt
for CFG in 1:2000:10; do
  rm -f /tmp/mc.trace
  MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=$CFG REPS=1 timeout 300 python tools/mchol_check.py > /dev/null 2>&1
  python tools/mchol_trace.py /tmp/mc.trace -2 0 > $O/mchol_trace_${CFG//:/_}.txt 2>&1
done
