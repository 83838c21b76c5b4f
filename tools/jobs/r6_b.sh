#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_b; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "adaptive or medium_configs" 2>&1 > $O/adaptive_tests.txt
grep -a "theta -\|passed\|failed\|Error" $O/adaptive_tests.txt | cut -c1-230 | head -150
