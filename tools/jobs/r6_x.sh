#!/bin/bash
export TMPDIR=/tmp
cd /root/repo; O=gpurun_out/r6_x; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "knife_edge or every_schedule" 2>&1 > $O/sweep.txt; grep -a "theta -\|passed\|failed\|Error\|band" $O/sweep.txt | cut -c1-200 | head -40
FUZZ_ONLY=87,900,1045 timeout 600 python tests/tools/fuzz_parity.py 1500 621 2>&1 | tail -8 | cut -c1-260
FUZZ_ONLY=169,528,588,883 timeout 600 python tests/tools/fuzz_parity.py 1500 622 2>&1 | tail -14 | cut -c1-260
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "mean or fuzz or random" 2>&1 | tail -4
