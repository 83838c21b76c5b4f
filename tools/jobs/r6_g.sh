#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_g; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "non_finite" 2>&1 | tail -60 > $O/t.txt; cat $O/t.txt | cut -c1-220
