#!/bin/bash
# gradient reduction: branch-free pairs, K^-1 requested up front, LDS reduction -- against the HEAD library
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3s; rm -rf $O; mkdir -p $O
for lib in build_ab/lib_head.so mogp_emulator_amd/libmogp_hip.so; do
  MOGP_LIB_PATH=$PWD/$lib REPS=6 timeout 300 python tools/kern_times.py 2>&1 | tee -a $O/kt.log | grep -E "fit|grad_reduce|kinv|trtri"
  MOGP_LIB_PATH=$PWD/$lib REPS=4 B=16 N=5000 D=20 M=2000 KERNEL=Matern52 timeout 300 python tools/kern_times.py 2>&1 | tee -a $O/kt.log | grep -E "fit|grad_reduce"
done
WHAT=grad REPS=6 timeout 600 python tools/ab.py "MOGP_LIB_PATH=$PWD/build_ab/lib_head.so" "" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grad or deriv or logpost or mean" 2>&1 | tail -3
