#!/bin/bash
# one-launch Cholesky, chain-bound launches (one workgroup per CU): kernel variant with __launch_bounds__(256, 1) -- spills in AGPRs, no scratch
export TMPDIR=/tmp
cd /root/repo
run() { echo "== $1"; env $1 WHAT=fit REPS=${2:-16} timeout 600 python tools/ab.py "MOGP_MC_SOLO=0" "" "MOGP_MC_SOLO=0" "" 2>&1 | tail -4 | cut -c1-130; }
run "B=8 N=2000 D=10 M=128"
run "B=4 N=2000 D=10 M=128"
run "B=1 N=2000 D=10 M=128"
run "B=2 N=5000 D=20 M=128 KERNEL=Matern52" 10
run "B=1 N=5000 D=20 M=128" 10
run "B=3 N=700 D=5 M=128"
