#!/bin/bash
# A/B of two library builds (.ab/<name>.so against the in-tree build): results + per-kernel times.  usage: REF=.ab/lib_x.so [SHAPES="B:n:d ..."] TAG=x bash tools/jobs/r4_libab.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/libab_${TAG:-head}; mkdir -p $O
{
WHAT=${WHAT:-fit,grad} REPS=${REPS:-8} timeout 900 python tools/ab.py "MOGP_LIB_PATH=/root/repo/$REF" "" "MOGP_LIB_PATH=/root/repo/$REF" ""
for shp in ${SHAPES:-64:2000:10}; do
  IFS=':' read -r b n d <<< "$shp"
  for lib in "/root/repo/$REF" ""; do
    MOGP_LIB_PATH=$lib B=$b N=$n D=$d M=${M:-2000} REPS=4 timeout 600 python tools/kern_times.py
  done
done
} 2>&1 | grep -v "^$" | tee $O/libab.txt
