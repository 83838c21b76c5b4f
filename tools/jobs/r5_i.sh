#!/bin/bash
# round 5: predict-phase kernels of library builds (LIBS; "intree" = the in-tree build) at C3 / C2 / 8-emulator / C4 shapes, optionally the suite
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5i_${TAG:-head}; rm -rf $O; mkdir -p $O
{
for rep in 1 2; do
for lib in $LIBS; do
  p=/root/repo/$lib; [ "$lib" = intree ] && p=""
  MOGP_LIB_PATH=$p B=64 N=2000 D=10 M=10000 REPS=4 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|cross_cov\|predict_var"
done; done
for lib in $LIBS; do
  p=/root/repo/$lib; [ "$lib" = intree ] && p=""
  MOGP_LIB_PATH=$p B=8 N=2000 D=10 M=10000 REPS=6 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|cross_cov\|predict_var"
  MOGP_LIB_PATH=$p B=1 N=2000 D=10 M=10000 REPS=6 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|cross_cov\|predict_var"
  MOGP_LIB_PATH=$p KERNEL=Matern52 B=16 N=5000 D=20 M=10000 REPS=3 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|cross_cov\|predict_var"
  MOGP_LIB_PATH=$p B=1 N=16000 D=8 M=10000 REPS=3 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|cross_cov\|predict_var"
done
first=$(echo $LIBS | cut -d' ' -f1)
WHAT=predict REPS=4 timeout 900 python tools/ab.py "MOGP_LIB_PATH=/root/repo/$first" ""
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/ab.txt
if [ -n "$SUITE" ]; then
  timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > $O/gpu_tests.txt; tail -15 $O/gpu_tests.txt
fi
