#!/bin/bash
# round 4: library A/B of the diagonal-block factorisation (column groups against the build in $REF) on chain-bound and full launches
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/c128ab_${TAG:-head}; mkdir -p $O
{
for shp in ${SHAPES:-1:2000:10 8:2000:10 64:2000:10 2:5000:20}; do
  IFS=':' read -r b n d <<< "$shp"
  echo "== $shp"
  B=$b N=$n D=$d WHAT=fit,grad REPS=${REPS:-15} timeout 600 python tools/ab.py "MOGP_LIB_PATH=/root/repo/$REF" "" "MOGP_LIB_PATH=/root/repo/$REF" ""
done
} 2>&1 | grep -v "^$" | tee $O/ab.txt
