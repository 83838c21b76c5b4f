#!/bin/bash
# round 4: number of chain-task row pairs (MOGP_MC_URG) in the launches with two workgroups per CU
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/urg_${TAG:-head}; mkdir -p $O
{
for shp in ${SHAPES:-16:2000:10 32:2000:10 64:2000:10 4:5000:20 1:16000:8}; do
  IFS=':' read -r b n d <<< "$shp"
  echo "== $shp"
  B=$b N=$n D=$d WHAT=fit REPS=${REPS:-12} timeout 900 python tools/ab.py "MOGP_MC_URG=0" "MOGP_MC_URG=1" "MOGP_MC_URG=2" "MOGP_MC_URG=0" "MOGP_MC_URG=2"
done
} 2>&1 | grep -v "^$" | tee $O/ab.txt
