#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_c; rm -rf $O; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -m gpu -q -x -k "bench or non_finite or backsolve or sharded or rccl or two_ranks or not_fit or predict" 2>&1 | tail -30 > $O/tests.txt
cat $O/tests.txt | cut -c1-250
