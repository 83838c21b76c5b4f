#!/bin/bash
# more randomised differential cases on the final kernels (three seeds in parallel), odd batches, stress shapes
export TMPDIR=/tmp
cd /root/repo
for sd in 311 312 313; do ( timeout 1100 python -W ignore tests/tools/fuzz_parity.py 1500 $sd 2>&1 | tail -4 ) > gpurun_out/r3m_fuzz_$sd.log & done
( timeout 1100 python -W ignore tests/tools/fuzz_parity.py 300 314 large 2>&1 | tail -4 ) > gpurun_out/r3m_fuzz_314.log &
wait
tail -n 4 gpurun_out/r3m_fuzz_31*.log
MOGP_MCHOL=0 timeout 600 python tests/tools/odd_batches.py 2>&1 | tail -6
timeout 600 python tests/tools/odd_batches.py 2>&1 | tail -6
