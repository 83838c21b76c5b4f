#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/k; export TMPDIR=/tmp; rm -f gpurun_out/k/rev.txt
for cfg in "64 2000" "8 2000" "16 5000" "2 5000"; do set -- $cfg
  echo "== B=$1 N=$2" >> gpurun_out/k/rev.txt
  B=$1 N=$2 WHAT=grad REPS=8 M=256 timeout 600 python tools/ab.py "MOGP_KINV_REV=0 MOGP_TRTRI_REV=0" "MOGP_KINV_REV=1 MOGP_TRTRI_REV=0" "MOGP_KINV_REV=0 MOGP_TRTRI_REV=1" "" >> gpurun_out/k/rev.txt 2>&1
done
cat gpurun_out/k/rev.txt
