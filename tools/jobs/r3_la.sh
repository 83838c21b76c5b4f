#!/bin/bash
# one-launch Cholesky: task orders MOGP_MC_LA = 1 (default) / 2 / 3
# (MOGP_MC_LA was an experiment of this job only: the alternative task orders are not in the tree, DESIGN.md section 5 list)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3la; rm -rf $O; mkdir -p $O
run() { echo "== $1"; env $1 WHAT=fit REPS=${2:-16} timeout 600 python tools/ab.py "" "MOGP_MC_LA=3" "MOGP_MC_LA=2" "" "MOGP_MC_LA=3" 2>&1 | tail -5 | cut -c1-110; }
( run "B=8 N=2000 D=10 M=128"
  run "B=4 N=2000 D=10 M=128"
  run "B=16 N=2000 D=10 M=128"
  run "B=64 N=2000 D=10 M=128" 12
  run "B=2 N=5000 D=20 M=128 KERNEL=Matern52" 10
  run "B=1 N=5000 D=20 M=128" 10 ) 2>&1 | tee $O/la.log
rm -f /tmp/mc.trace; MOGP_MC_LA=3 MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=8:2000:10 REPS=1 timeout 300 python tools/mchol_check.py > /dev/null 2>&1
python tools/mchol_trace.py /tmp/mc.trace -2 0 > $O/trace_8_la3.txt 2>&1; cut -c1-150 $O/trace_8_la3.txt | sed -n 2,26p
