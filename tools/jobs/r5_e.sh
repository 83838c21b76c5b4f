#!/bin/bash
# round 5: tail overlap of the back substitution: A/B (MOGP_BS_OVERLAP=0 / 1) at several batch sizes, then the GPU suite.  usage: TAG=x [SUITE=1] bash tools/jobs/r5_e.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5e_${TAG:-head}; rm -rf $O; mkdir -p $O
{
for shp in "64 2000 10" "32 2000 10" "16 2000 10" "8 2000 10" "1 2000 10" "16 5000 20" "1 16000 8"; do
  set -- $shp
  echo "== $shp"
  B=$1 N=$2 D=$3 WHAT=fit REPS=12 timeout 600 python tools/ab.py "MOGP_BS_OVERLAP=0" "MOGP_BS_OVERLAP=1" "MOGP_BS_OVERLAP=0" "MOGP_BS_OVERLAP=1"
done
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/ab.txt
if [ -n "$SUITE" ]; then
  timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > $O/gpu_tests.txt; tail -15 $O/gpu_tests.txt
fi
