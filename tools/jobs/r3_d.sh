#!/bin/bash
# round 3, run D: one-launch Cholesky as default -- full GPU suite, odd batches, stress shapes, timing sweep
export TMPDIR=/tmp
cd /root/repo
timeout 1700 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 > gpurun_out/r3d_tests.txt
tail -14 gpurun_out/r3d_tests.txt
timeout 600 python tests/tools/odd_batches.py 2>&1 | tail -8
timeout 600 python tests/tools/stress_shapes.py 2>&1 | tail -8
MOGP_MC_SPIN=400000 timeout 600 python tools/mchol_check.py 2>&1 | tail -12 | cut -c1-250
