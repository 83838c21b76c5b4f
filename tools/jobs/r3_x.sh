#!/bin/bash
# case 308 of fuzz seed 323 (pivot, two repeated points): current library, the library of the start of this session (b0f0bec), forced
# multi-launch schedules -- is the difference from the oracle new?
export TMPDIR=/tmp
cd /root/repo
for cfg in "X=1" "MOGP_LIB_PATH=$PWD/build_ab/lib_head.so" "MOGP_MCHOL=0"; do
  echo "== $cfg"; env $cfg FUZZ_ONLY=308 timeout 900 python -W ignore tests/tools/fuzz_parity.py 1500 323 2>&1 | grep -E "MISMATCH|EXCEPTION|cases" | cut -c1-120
done
