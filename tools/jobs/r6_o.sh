#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_o; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "objective_with or fit_GP_MAP or gradient" 2>&1 | tail -15 | cut -c1-200
