#!/bin/bash
# round 6, last job: the GPU suite, the odd-batch / shape stress checkers and the multi-start timing on the final HEAD
export TMPDIR=/tmp
cd /root/repo; O=gpurun_out/r6_w; rm -rf $O; mkdir -p $O
cp mogp_emulator_amd/libmogp_hip.build $O/build_commit.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -12 > $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/gpu_tests.txt
timeout 900 python tests/tools/odd_batches.py 2>&1 | tail -5 > $O/odd_batches.txt; tail -2 $O/odd_batches.txt
timeout 900 python tests/tools/stress_shapes.py 2>&1 | tail -5 > $O/stress_shapes.txt; tail -2 $O/stress_shapes.txt
