#!/bin/bash
# randomised differential cases on the final kernels of round 3 (new seeds; one seed with the lock-step predictive variance), odd batches
export TMPDIR=/tmp
cd /root/repo
for sd in 321 322; do ( timeout 1100 python -W ignore tests/tools/fuzz_parity.py 1500 $sd 2>&1 | tail -4 ) > gpurun_out/r3v_fuzz_$sd.log & done
( MOGP_PV_SYNC=1000 timeout 1100 python -W ignore tests/tools/fuzz_parity.py 1500 323 2>&1 | tail -4 ) > gpurun_out/r3v_fuzz_323_pvsync.log &
( timeout 1100 python -W ignore tests/tools/fuzz_parity.py 300 324 large 2>&1 | tail -4 ) > gpurun_out/r3v_fuzz_324_large.log &
wait
tail -n 4 gpurun_out/r3v_fuzz_32*.log
timeout 600 python tests/tools/odd_batches.py 2>&1 | tail -6
