#!/bin/bash
# round 4: small-launch variants (32-deep GEMM steps in one-per-CU Cholesky launches, 64 x 64 K^-1 tiles) against the previous build
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/small_${TAG:-head}; mkdir -p $O
{
for shp in 1:2000:10 4:2000:10 8:2000:10 2:5000:20 3:700:5; do
  IFS=':' read -r b n d <<< "$shp"
  echo "== $shp"
  B=$b N=$n D=$d WHAT=fit,grad REPS=15 timeout 600 python tools/ab.py "MOGP_MC_BK32=0 MOGP_KINV_WT=4 MOGP_MC_LATE=0" "MOGP_MC_BK32=1 MOGP_KINV_WT=4 MOGP_MC_LATE=0" "MOGP_MC_BK32=1 MOGP_KINV_WT=4 MOGP_MC_LATE=1" "MOGP_MC_BK32=0 MOGP_KINV_WT=4 MOGP_MC_LATE=1" "MOGP_MC_BK32=1 MOGP_KINV_WT=2 MOGP_MC_LATE=1" ""
done
} 2>&1 | grep -v "^$" | tee $O/small.txt
