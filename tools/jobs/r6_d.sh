#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_d; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python tools/big_batch.py > $O/big_batch.txt 2>&1; cat $O/big_batch.txt
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -q -x -k "eight" 2>&1 | tail -5
