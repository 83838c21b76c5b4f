#!/bin/bash
# round 5: env-switch A/B with per-kernel times of the predict phase.  usage: CFGS="A=0;A=1" TAG=x bash tools/jobs/r5_j.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5j_${TAG:-head}; rm -rf $O; mkdir -p $O
IFS=';' read -ra CF <<< "${CFGS}"
{
for rep in 1 2; do for cfg in "${CF[@]}"; do
  echo "[$cfg]"; env $cfg B=64 N=2000 D=10 M=10000 REPS=4 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|cross_cov\|predict_var"
done; done
for cfg in "${CF[@]}"; do
  echo "[$cfg] 8 x n=2000, C2, C5"; env $cfg B=8 N=2000 D=10 M=10000 REPS=6 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|cross_cov\|predict_var"
  env $cfg B=1 N=2000 D=10 M=10000 REPS=6 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|cross_cov\|predict_var"
  env $cfg B=1 N=16000 D=8 M=10000 REPS=3 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|cross_cov\|predict_var"
done
WHAT=predict REPS=4 timeout 900 python tools/ab.py "${CF[@]}" "${CF[@]}"
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/ab.txt
