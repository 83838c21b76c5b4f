#!/bin/bash
# round 4: GPU suite only (+ smoke); usage: TAG=x bash tools/jobs/r4_tests.sh
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/tests_${TAG:-head}; rm -rf $O; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -20 > $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
