#!/bin/bash
# round 5: panel-layout K* and SOLO -C init against .ab/lib_head.so, then the GPU suite
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5h_${TAG:-head}; rm -rf $O; mkdir -p $O
{
for lib in .ab/lib_head.so intree .ab/lib_head.so intree; do
  p=/root/repo/$lib; [ "$lib" = intree ] && p=""
  MOGP_LIB_PATH=$p B=64 N=2000 D=10 M=10000 REPS=4 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|mchol\|cross_cov\|predict_var"
done
for shp in "8 2000 10" "1 2000 10"; do
  set -- $shp
  for lib in .ab/lib_head.so intree .ab/lib_head.so intree; do
    p=/root/repo/$lib; [ "$lib" = intree ] && p=""
    MOGP_LIB_PATH=$p B=$1 N=$2 D=$3 M=10000 REPS=6 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|mchol\|cross_cov\|predict_var"
  done
done
for lib in .ab/lib_head.so intree; do
  p=/root/repo/$lib; [ "$lib" = intree ] && p=""
  MOGP_LIB_PATH=$p KERNEL=Matern52 B=16 N=5000 D=20 M=10000 REPS=3 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|mchol\|cross_cov\|predict_var"
done
WHAT=predict REPS=4 timeout 900 python tools/ab.py "MOGP_LIB_PATH=/root/repo/.ab/lib_head.so" ""
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/ab.txt
if [ -n "$SUITE" ]; then
  timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > $O/gpu_tests.txt; tail -15 $O/gpu_tests.txt
fi
