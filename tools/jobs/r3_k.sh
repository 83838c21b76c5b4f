#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/k; export TMPDIR=/tmp; rm -f gpurun_out/k/misc.txt
for cfg in "1 2000 10000" "1 500 10000" "4 2000 10000" "64 2000 100000" "512 500 1000" "2000 100 1000"; do set -- $cfg
  echo "== B=$1 N=$2 M=$3" >> gpurun_out/k/misc.txt
  B=$1 N=$2 M=$3 REPS=5 timeout 600 python tools/ab.py "" 2>&1 | tail -n 1 >> gpurun_out/k/misc.txt
done
cat gpurun_out/k/misc.txt
