#!/bin/bash
# round 5: predict_deriv (prefetching form) against .ab/lib_head.so, then the GPU suite
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5k_${TAG:-head}; rm -rf $O; mkdir -p $O
{
for lib in .ab/lib_head.so intree .ab/lib_head.so intree; do
  p=/root/repo/$lib; [ "$lib" = intree ] && p=""
  MOGP_LIB_PATH=$p timeout 300 python tools/deriv_timing.py
done
for lib in .ab/lib_head.so intree; do
  p=/root/repo/$lib; [ "$lib" = intree ] && p=""
  MOGP_LIB_PATH=$p B=16 N=5000 D=20 M=10000 KERNEL=Matern52 timeout 300 python tools/deriv_timing.py
  MOGP_LIB_PATH=$p B=1 N=2000 D=10 M=10000 timeout 300 python tools/deriv_timing.py
done
DERIV=1 B=64 N=2000 D=10 M=10000 REPS=3 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|cross_cov\|predict_var\|predict_deriv"
DERIV=1 B=16 N=5000 D=20 M=10000 KERNEL=Matern52 REPS=2 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|cross_cov\|predict_var\|predict_deriv"
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/ab.txt
if [ -n "$SUITE" ]; then
  timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > $O/gpu_tests.txt; tail -15 $O/gpu_tests.txt
fi
