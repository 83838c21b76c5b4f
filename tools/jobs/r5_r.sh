#!/bin/bash
# round 5: task order of the last block columns of the one-launch Cholesky (MOGP_MC_ORDER, MOGP_MC_ORDER_FROM)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5r_${TAG:-head}; rm -rf $O; mkdir -p $O
{
for shp in ${SHAPES:-"64 2000 10" "32 2000 10" "16 5000 20"}; do
  set -- $shp
  for cfg in "MOGP_MC_ORDER=0" "MOGP_MC_ORDER=1" "MOGP_MC_ORDER=3" "MOGP_MC_ORDER=4" "MOGP_MC_ORDER=7" "MOGP_MC_ORDER=5" "MOGP_MC_ORDER=0" "MOGP_MC_ORDER=3 MOGP_MC_ORDER_FROM=6" "MOGP_MC_ORDER=7 MOGP_MC_ORDER_FROM=3" "MOGP_MC_ORDER=1 MOGP_MC_ORDER_FROM=100"; do
    env $cfg B=$1 N=$2 D=$3 REPS=10 timeout 300 python tools/mchol_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[$cfg] /"
  done
done
} 2>&1 | tee $O/sweep.txt
