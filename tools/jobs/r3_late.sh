#!/bin/bash
# one-launch Cholesky, chain-bound launches: a second workgroup per CU that sleeps until block column pct/100 K (MOGP_MC_LATE, experiment)
# (MOGP_MC_LATE was an experiment of this job only: not in the tree, DESIGN.md section 5 list)
export TMPDIR=/tmp
cd /root/repo
run() { echo "== $1"; env $1 WHAT=fit REPS=${2:-16} timeout 600 python tools/ab.py "" "MOGP_MC_LATE=30" "MOGP_MC_LATE=50" "MOGP_MC_LATE=70" "" "MOGP_MC_LATE=50" 2>&1 | tail -6 | cut -c1-110; }
run "B=8 N=2000 D=10 M=128"
run "B=4 N=2000 D=10 M=128"
run "B=2 N=5000 D=20 M=128 KERNEL=Matern52" 10
run "B=1 N=5000 D=20 M=128" 10
