#!/bin/bash
# after the predictive-variance / gradient-reduction changes: kernel times against HEAD, new tests, then the whole GPU suite
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3t; rm -rf $O; mkdir -p $O
for lib in build_ab/lib_head.so mogp_emulator_amd/libmogp_hip.so; do
  MOGP_LIB_PATH=$PWD/$lib REPS=6 timeout 300 python tools/kern_times.py 2>&1 | tee -a $O/kt.log | grep -E "fit|grad_reduce|predict_var"
  MOGP_LIB_PATH=$PWD/$lib REPS=4 B=16 N=5000 D=20 M=2000 KERNEL=Matern52 timeout 300 python tools/kern_times.py 2>&1 | tee -a $O/kt.log | grep -E "fit|grad_reduce|predict_var"
done
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lockstep or PV_" 2>&1 | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16 > $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt
