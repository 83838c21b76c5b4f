#!/bin/bash
# round 5: per-kernel times of several library builds (.ab/*.so, "" = in-tree) at the C3 shape (+ C4 for the last two), then the GPU suite.
# usage: LIBS=".ab/a.so .ab/b.so intree" TAG=x [SUITE=1] bash tools/jobs/r5_b.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5b_${TAG:-head}; rm -rf $O; mkdir -p $O
{
for lib in $LIBS; do
  p=/root/repo/$lib; [ "$lib" = intree ] && p=""
  MOGP_LIB_PATH=$p B=64 N=2000 D=10 M=${M:-10000} REPS=4 timeout 600 python tools/kern_times.py
done
for lib in $C4LIBS; do
  p=/root/repo/$lib; [ "$lib" = intree ] && p=""
  MOGP_LIB_PATH=$p KERNEL=Matern52 B=16 N=5000 D=20 M=2000 REPS=3 timeout 600 python tools/kern_times.py
done
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/kern.txt
if [ -n "$SUITE" ]; then
  timeout 2700 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -30 > $O/gpu_tests.txt; tail -12 $O/gpu_tests.txt
fi
