#!/bin/bash
# one-launch Cholesky: long-K GEMM of the band rows split (H tasks park the accumulators) -- MOGP_MC_SPLIT=0 / default / the start-of-session library
# (MOGP_MC_SPLIT / the H tasks were an experiment of this job only: not in the tree, DESIGN.md section 5 list)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3sp; rm -rf $O; mkdir -p $O
H="MOGP_LIB_PATH=$PWD/build_ab/lib_head.so"
run() { echo "== $1"; env $1 WHAT=fit REPS=${2:-16} timeout 600 python tools/ab.py "MOGP_MC_SPLIT=0" "" "$H" "MOGP_MC_SPLIT=0" "" 2>&1 | tail -5 | cut -c1-150; }
( run "B=8 N=2000 D=10 M=128"
  run "B=4 N=2000 D=10 M=128"
  run "B=2 N=5000 D=20 M=128 KERNEL=Matern52" 10
  run "B=1 N=5000 D=20 M=128" 10
  run "B=16 N=2000 D=10 M=128"
  run "B=64 N=2000 D=10 M=128" 12
  run "B=1 N=16000 D=8 M=128" 4 ) 2>&1 | tee $O/sp.log
rm -f /tmp/mc.trace; MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=8:2000:10 REPS=1 timeout 300 python tools/mchol_check.py 2>&1 | tail -3
python tools/mchol_trace.py /tmp/mc.trace -2 0 > $O/trace_8_split.txt 2>&1; cut -c1-150 $O/trace_8_split.txt | sed -n 2,28p
