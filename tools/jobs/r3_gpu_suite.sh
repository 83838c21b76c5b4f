#!/bin/bash
# the GPU suite + smoke() on the current tree; usage: TAG=<commit> bash tools/jobs/r3_gpu_suite.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/suite_${TAG:-head}; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 2>&1 | tail -22 > $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
