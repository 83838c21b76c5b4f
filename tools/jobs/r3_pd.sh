#!/bin/bash
# one-launch Cholesky, one-workgroup-per-CU build: operand prefetch 8 k-steps ahead instead of 4 -- against commit fe3ce50
# (MOGP_PD_SOLO was an experiment of this job only: not in the tree, DESIGN.md section 5 list)
export TMPDIR=/tmp
cd /root/repo
P="MOGP_LIB_PATH=$PWD/build_ab/lib_prev.so"
run() { echo "== $1"; env $1 WHAT=fit REPS=${2:-16} timeout 600 python tools/ab.py "$P" "" "$P" "" 2>&1 | tail -4 | cut -c1-130; }
run "B=8 N=2000 D=10 M=128"
run "B=4 N=2000 D=10 M=128"
run "B=2 N=5000 D=20 M=128 KERNEL=Matern52" 10
run "B=1 N=5000 D=20 M=128" 10
