#!/bin/bash
# are the pivot / two-repeated-points differences of the round-3 fuzz seeds new?  same seeds on the round-2 library
export TMPDIR=/tmp
cd /root/repo
( MOGP_LIB_PATH=$PWD/build_ab/lib_r2.so timeout 900 python -W ignore tests/tools/fuzz_parity.py 230 313 2>&1 | tail -8 ) > gpurun_out/r3n_r2_313.log &
( MOGP_LIB_PATH=$PWD/build_ab/lib_r2.so timeout 1100 python -W ignore tests/tools/fuzz_parity.py 1030 311 2>&1 | tail -4 ) > gpurun_out/r3n_r2_311.log &
( timeout 900 python -W ignore tests/tools/fuzz_parity.py 230 313 2>&1 | tail -8 ) > gpurun_out/r3n_r3_313.log &
( timeout 1100 python -W ignore tests/tools/fuzz_parity.py 300 314 large 2>&1 | tail -4 ) > gpurun_out/r3n_r3_314.log &
wait
tail -n 8 gpurun_out/r3n_*.log
