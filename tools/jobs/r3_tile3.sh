#!/bin/bash
# one-workgroup-per-CU build: chain tasks through the LDS tile hand-off as well (MOGP_MC_TILE=2) against bulk tasks only (1, default)
# (MOGP_MC_TILE=2 was an experiment of this job only: not in the tree, DESIGN.md section 5 list)
export TMPDIR=/tmp
cd /root/repo
run() { echo "== $1"; env $1 WHAT=fit REPS=${2:-16} timeout 600 python tools/ab.py "" "MOGP_MC_TILE=2" "" "MOGP_MC_TILE=2" 2>&1 | tail -4 | cut -c1-130; }
run "B=8 N=2000 D=10 M=128"
run "B=4 N=2000 D=10 M=128"
run "B=2 N=5000 D=20 M=128 KERNEL=Matern52" 10
run "B=1 N=5000 D=20 M=128" 10
