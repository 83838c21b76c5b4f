#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/k; export TMPDIR=/tmp; rm -f gpurun_out/k/ls.txt
for cfg in "8 2000 10 10" "8 2000 10 200" "8 500 6 200"; do set -- $cfg
  echo "== B=$1 N=$2 D=$3 MAXITER=$4" >> gpurun_out/k/ls.txt
  B=$1 N=$2 D=$3 MAXITER=$4 timeout 900 python tools/fitmap_timing.py 2>&1 | tail -n 2 >> gpurun_out/k/ls.txt
done
cat gpurun_out/k/ls.txt
