#!/bin/bash
# one-launch Cholesky, chain-bound regimes: workgroups per CU / parking revisited on the final kernels
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3w; rm -rf $O; mkdir -p $O
run() { echo "== $1"; env $1 WHAT=fit REPS=16 timeout 600 python tools/ab.py "" "MOGP_MC_WGS=2 MOGP_MC_PARK=1" "MOGP_MC_WGS=2 MOGP_MC_PARK=0" "MOGP_MC_WGS=1" "" 2>&1 | tail -5 | cut -c1-120; }
( run "B=8 N=2000 D=10 M=128"
  run "B=4 N=2000 D=10 M=128"
  run "B=16 N=2000 D=10 M=128"
  run "B=2 N=5000 D=20 M=128 KERNEL=Matern52"
  run "B=1 N=5000 D=20 M=128"
  run "B=1 N=16000 D=8 M=128" ) 2>&1 | tee $O/wgs.log
# the pivot / two-repeated-points mismatches of seed 323: same seed on the round-2 library and on HEAD without the lock-step switch
( MOGP_LIB_PATH=$PWD/build_ab/lib_r2.so timeout 1200 python -W ignore tests/tools/fuzz_parity.py 1500 323 2>&1 | grep -E "MISMATCH|cases" ) > gpurun_out/r3w_fuzz_323_r2lib.log &
( timeout 1200 python -W ignore tests/tools/fuzz_parity.py 1500 323 2>&1 | grep -E "MISMATCH|cases" ) > gpurun_out/r3w_fuzz_323_head.log &
wait
tail -n 12 gpurun_out/r3w_fuzz_323_*.log
