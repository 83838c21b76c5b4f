#!/bin/bash
# round 4: back substitution with the column entries of a diagonal block in registers before the chain starts, against the build in $REF
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/bs2_${TAG:-head}; mkdir -p $O
{
for shp in ${SHAPES:-1:2000:10 8:2000:10 16:2000:10 64:2000:10 2:5000:20 1:16000:8 3:700:5 1:129:3}; do
  IFS=':' read -r b n d <<< "$shp"
  echo "== $shp"
  B=$b N=$n D=$d WHAT=fit,predict M=500 REPS=${REPS:-15} timeout 600 python tools/ab.py "MOGP_LIB_PATH=/root/repo/$REF" "" "MOGP_LIB_PATH=/root/repo/$REF" ""
done
} 2>&1 | grep -v "^$" | tee $O/ab.txt
