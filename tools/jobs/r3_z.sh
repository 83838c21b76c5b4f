#!/bin/bash
# the pivot cases of six fuzz seeds (the three of the earlier round-3 runs, three newer ones) with the blocked-dpstrf tail
export TMPDIR=/tmp
cd /root/repo
for sd in 311 312 313 321 323 325; do ( FUZZ_NUGGET=pivot timeout 1500 python -W ignore tests/tools/fuzz_parity.py 1500 $sd 2>&1 | grep -E "MISMATCH|EXCEPTION|cases|repeated" | cut -c1-200 ) > gpurun_out/r3z_fuzz_$sd.log & done
( FUZZ_NUGGET=pivot timeout 1500 python -W ignore tests/tools/fuzz_parity.py 300 314 large 2>&1 | grep -E "MISMATCH|EXCEPTION|cases|repeated" | cut -c1-200 ) > gpurun_out/r3z_fuzz_314_large.log &
wait
tail -n 6 gpurun_out/r3z_fuzz_*.log
