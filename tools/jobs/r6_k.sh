#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_k; rm -rf $O; mkdir -p $O; cd $R
{ for rep in 1 2; do timeout 600 python tools/cov_ab.py; MOGP_LIB_PATH=$R/.ab/libmogp_unscaled.so timeout 600 python tools/cov_ab.py; done; } 2>&1 | grep -v amdgpu > $O/cov_ab.txt; cat $O/cov_ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullbatch.py -m gpu -q -x -k "golden or medium or fixture or grid11 or c4 or c2_full or kernel_objects or c3_full or matern" 2>&1 | tail -5
