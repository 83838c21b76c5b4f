#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/k; export TMPDIR=/tmp; rm -f gpurun_out/k/rep.txt
for cfg in "64 2000 10" "32 2000 10" "16 5000 20" "4 5000 20"; do set -- $cfg
  echo "== B=$1 N=$2 D=$3" >> gpurun_out/k/rep.txt
  B=$1 N=$2 D=$3 timeout 900 python tools/fitmap_timing.py 2>&1 | tail -n 2 >> gpurun_out/k/rep.txt
done
cat gpurun_out/k/rep.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fit_GP_MAP or starts" 2>&1 | tail -n 2
