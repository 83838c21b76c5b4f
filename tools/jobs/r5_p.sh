#!/bin/bash
# round 5: dispatch order of the one-launch back substitution (emulators per group, MOGP_BS_GROUP) by batch size
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5p_${TAG:-head}; rm -rf $O; mkdir -p $O
{
for shp in ${SHAPES:-"64 2000 10" "32 2000 10" "16 5000 20" "128 1000 5"}; do
  set -- $shp
  for g in 0 32 16 8 4 1 0 16; do
    echo "[MOGP_BS_GROUP=$g] $shp"
    MOGP_BS_GROUP=$g B=$1 N=$2 D=$3 M=${M:-256} REPS=${REPS:-8} timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|backsolve"
  done
done
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/ab.txt
