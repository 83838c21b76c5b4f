#!/bin/bash
# one-launch Cholesky: bulk tasks hand C - acc to the panel solve through LDS (default) instead of through global memory (MOGP_MC_TILE=0)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3tile; rm -rf $O; mkdir -p $O
run() { echo "== $1"; env $1 WHAT=fit REPS=${2:-16} timeout 600 python tools/ab.py "MOGP_MC_TILE=0" "" "MOGP_MC_TILE=0" "" 2>&1 | tail -4 | cut -c1-150; }
( run "B=64 N=2000 D=10 M=128" 12
  run "B=8 N=2000 D=10 M=128"
  run "B=16 N=2000 D=10 M=128"
  run "B=16 N=5000 D=20 M=128 KERNEL=Matern52" 6
  run "B=2 N=5000 D=20 M=128 KERNEL=Matern52" 10
  run "B=1 N=16000 D=8 M=128" 4
  run "B=3 N=700 D=5 M=128"
  run "B=64 N=1000 D=10 M=128" ) 2>&1 | tee $O/tile.log
rm -f /tmp/mc.trace; MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=64:2000:10 REPS=1 timeout 300 python tools/mchol_check.py 2>&1 | tail -2
python tools/mchol_trace.py /tmp/mc.trace -2 0 2>&1 | tail -8 | cut -c1-200
