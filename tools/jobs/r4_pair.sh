#!/bin/bash
# round 4: paired 128 x 128 bulk tasks of the one-launch Cholesky vs 64 x 128, and the traffic-free A/B (MOGP_MC_NOTRAFFIC)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/pair_${TAG:-head}; rm -rf $O; mkdir -p $O
{
echo "== results: paired vs unpaired must be bit-identical (d_f = d_g = 0)"
WHAT=fit,grad REPS=10 timeout 600 python tools/ab.py "MOGP_MC_PAIR=0" "MOGP_MC_PAIR=1"
B=32 WHAT=fit REPS=10 timeout 600 python tools/ab.py "MOGP_MC_PAIR=0" "MOGP_MC_PAIR=1"
echo "== mchol kernel time"
for cfg in "MOGP_MC_PAIR=0" "MOGP_MC_PAIR=1" "MOGP_MC_PAIR=0 MOGP_MC_NOTRAFFIC=1" "MOGP_MC_PAIR=1 MOGP_MC_NOTRAFFIC=1"; do
  env $cfg timeout 300 python tools/mchol_time.py
done
for cfg in "MOGP_MC_PAIR=0" "MOGP_MC_PAIR=1"; do
  env $cfg B=32 timeout 300 python tools/mchol_time.py
  env $cfg B=16 timeout 300 python tools/mchol_time.py
  env $cfg B=16 N=5000 D=20 KERNEL=Matern52 REPS=5 timeout 300 python tools/mchol_time.py
  env $cfg B=1 N=16000 D=8 REPS=4 timeout 300 python tools/mchol_time.py
  env $cfg B=4 N=5000 D=20 REPS=5 timeout 300 python tools/mchol_time.py
done
} 2>&1 | grep -v "^$" | tee $O/pair_ab.txt
