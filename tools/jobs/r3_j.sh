#!/bin/bash
# round 3: long randomised differential runs and leak checks on the new default (one-launch Cholesky)
export TMPDIR=/tmp
cd /root/repo
( timeout 900 python -W ignore tests/tools/fuzz_parity.py 1500 301 2>&1 | tail -4 ) > gpurun_out/r3j_fuzz_a.log &
( timeout 900 python -W ignore tests/tools/fuzz_parity.py 250 302 large 2>&1 | tail -4 ) > gpurun_out/r3j_fuzz_b.log &
wait
cat gpurun_out/r3j_fuzz_a.log gpurun_out/r3j_fuzz_b.log
( MOGP_CHOL=mchol MOGP_MC_WGS=2 MOGP_MC_PARK=1 timeout 900 python -W ignore tests/tools/fuzz_parity.py 150 303 large 2>&1 | tail -3 ) > gpurun_out/r3j_fuzz_c.log
cat gpurun_out/r3j_fuzz_c.log
timeout 600 python tools/leak_check.py 2>&1 | tail -4
timeout 600 python tools/leak_check_pivot.py 2>&1 | tail -3
