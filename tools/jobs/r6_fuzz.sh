#!/bin/bash
# round 6: the randomised differential test on the round-6 kernels (new seeds)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/fuzz_${TAG:-head}; rm -rf $O; mkdir -p $O
for seed in ${SEEDS:-611 612}; do
  timeout 1500 python tests/tools/fuzz_parity.py 1500 $seed 2>&1 | tail -12 > $O/fuzz_$seed.txt; tail -3 $O/fuzz_$seed.txt
done
timeout 1500 python tests/tools/fuzz_parity.py 300 ${LSEED:-613} large 2>&1 | tail -12 > $O/fuzz_large.txt; tail -3 $O/fuzz_large.txt
MOGP_MC_WGS=2 timeout 900 python tests/tools/fuzz_parity.py 150 ${LSEED2:-614} large 2>&1 | tail -12 > $O/fuzz_large_wgs2.txt; tail -3 $O/fuzz_large_wgs2.txt
