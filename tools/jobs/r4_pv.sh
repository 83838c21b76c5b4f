#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/pv_${TAG:-head}; mkdir -p $O
{
for b in 1 2 3 4 8; do
  B=$b WHAT=predict REPS=10 timeout 600 python tools/ab.py "MOGP_PV_SINGLE=0" "MOGP_PV_SINGLE=1" ""
done
B=1 N=5000 D=20 WHAT=predict REPS=5 timeout 600 python tools/ab.py "MOGP_PV_SINGLE=0" "MOGP_PV_SINGLE=1" ""
B=1 N=700 D=5 M=3000 WHAT=predict REPS=10 timeout 600 python tools/ab.py "MOGP_PV_SINGLE=0" "MOGP_PV_SINGLE=1" ""
} 2>&1 | grep -v "^$" | tee $O/pv.txt
