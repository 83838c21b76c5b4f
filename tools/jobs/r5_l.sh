#!/bin/bash
# round 5: env-switch A/B of the fit phase (mchol) at throughput-bound shapes, then the suite.  usage: CFGS="A=0;A=1" TAG=x [SUITE=1] bash tools/jobs/r5_l.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5l_${TAG:-head}; rm -rf $O; mkdir -p $O
IFS=';' read -ra CF <<< "${CFGS}"
{
for shp in "64 2000 10" "32 2000 10" "16 5000 20" "1 16000 8" "16 2000 10"; do
  set -- $shp
  for rep in 1 2; do for cfg in "${CF[@]}"; do
    echo "[$cfg] $shp"; env $cfg B=$1 N=$2 D=$3 M=256 REPS=8 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|mchol"
  done; done
done
WHAT=fit,grad REPS=6 timeout 900 python tools/ab.py "${CF[@]}" "${CF[@]}"
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/ab.txt
if [ -n "$SUITE" ]; then
  timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > $O/gpu_tests.txt; tail -15 $O/gpu_tests.txt
fi
