#!/bin/bash
# round 5: GEMM main-loop probe, per-kernel times of the in-tree library, then the GPU suite.  usage: TAG=x [SUITE=1] bash tools/jobs/r5_c.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5c_${TAG:-head}; rm -rf $O; mkdir -p $O
{
for args in "64 40" "8 200" "4 300"; do echo "== gemm_loop_probe $args"; timeout 120 tools/gemm_loop_probe.bin $args; done
B=64 N=2000 D=10 M=10000 REPS=4 timeout 600 python tools/kern_times.py
B=8 N=2000 D=10 M=2000 REPS=6 timeout 600 python tools/kern_times.py
B=1 N=2000 D=10 M=2000 REPS=6 timeout 600 python tools/kern_times.py
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/probe.txt
if [ -n "$SUITE" ]; then
  timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > $O/gpu_tests.txt; tail -15 $O/gpu_tests.txt
fi
