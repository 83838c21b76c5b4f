#!/bin/bash
# round 5: A/B of the in-tree library against .ab/<REF> (results + per-kernel times at C3 and C4 shapes), then the GPU suite.
# usage: REF=.ab/lib_r4.so TAG=x [SUITE=1] bash tools/jobs/r5_a.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5a_${TAG:-head}; rm -rf $O; mkdir -p $O
{
WHAT=fit,grad,predict REPS=6 timeout 900 python tools/ab.py "MOGP_LIB_PATH=/root/repo/$REF" "" "MOGP_LIB_PATH=/root/repo/$REF" ""
for lib in "/root/repo/$REF" ""; do
  MOGP_LIB_PATH=$lib B=64 N=2000 D=10 M=10000 REPS=4 timeout 600 python tools/kern_times.py
done
for lib in "/root/repo/$REF" ""; do
  MOGP_LIB_PATH=$lib KERNEL=Matern52 B=16 N=5000 D=20 M=2000 REPS=3 timeout 600 python tools/kern_times.py
done
for lib in "/root/repo/$REF" ""; do
  MOGP_LIB_PATH=$lib B=8 N=2000 D=10 M=2000 REPS=6 timeout 600 python tools/kern_times.py
done
} 2>&1 | grep -v "^$" | tee $O/libab.txt
if [ -n "$SUITE" ]; then
  timeout 2700 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16 > $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt
fi
