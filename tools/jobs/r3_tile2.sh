#!/bin/bash
# one-launch Cholesky: the chain tasks also hand their tile to the (pipelined) panel solve through LDS -- against commit a9ab628 (bulk tasks only)
# (the chain-task variant was an experiment of this job only: not in the tree, DESIGN.md section 5 list)
export TMPDIR=/tmp
cd /root/repo
P="MOGP_LIB_PATH=$PWD/build_ab/lib_prev.so"
run() { echo "== $1"; env $1 WHAT=fit REPS=${2:-16} timeout 600 python tools/ab.py "$P" "" "$P" "" 2>&1 | tail -4 | cut -c1-150; }
run "B=8 N=2000 D=10 M=128"
run "B=4 N=2000 D=10 M=128"
run "B=2 N=5000 D=20 M=128 KERNEL=Matern52" 10
run "B=1 N=5000 D=20 M=128" 10
run "B=16 N=2000 D=10 M=128"
run "B=64 N=2000 D=10 M=128" 12
run "B=1 N=16000 D=8 M=128" 4
run "B=3 N=700 D=5 M=128"
