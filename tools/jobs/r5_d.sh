#!/bin/bash
# round 5: GEMM main-loop probe + the one-launch Cholesky with each main loop (library builds in .ab/), then the GPU suite on the in-tree build.
# usage: LIBS=".ab/a.so intree" TAG=x [SUITE=1] bash tools/jobs/r5_d.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5d_${TAG:-head}; rm -rf $O; mkdir -p $O
{
for args in "64 40" "8 200"; do echo "== gemm_loop_probe $args"; timeout 120 tools/gemm_loop_probe.bin $args; done
for shp in "64 2000 10" "8 2000 10" "1 2000 10" "16 2000 10" "32 2000 10"; do
  set -- $shp
  for lib in $LIBS; do
    p=/root/repo/$lib; [ "$lib" = intree ] && p=""
    MOGP_LIB_PATH=$p B=$1 N=$2 D=$3 M=1000 REPS=6 timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|mchol"
  done
done
WHAT=fit,grad REPS=6 timeout 900 python tools/ab.py "MOGP_LIB_PATH=/root/repo/.ab/lib_pf.so" ""
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/probe.txt
if [ -n "$SUITE" ]; then
  timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > $O/gpu_tests.txt; tail -15 $O/gpu_tests.txt
fi
