#!/bin/bash
# round 5: the regime rules of the one-launch Cholesky re-measured on the round-5 kernels: workgroups per CU x parking x chain-task rows, by batch size
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5m_${TAG:-head}; rm -rf $O; mkdir -p $O
{
for B in 4 8 12 16 24 32; do
  for cfg in "MOGP_MC_WGS=1" "MOGP_MC_WGS=2 MOGP_MC_PARK=1" "MOGP_MC_WGS=2 MOGP_MC_PARK=0" "MOGP_MC_WGS=1 MOGP_MC_URG=1" "MOGP_MC_WGS=1 MOGP_MC_URG=0" "MOGP_MC_WGS=2 MOGP_MC_PARK=1 MOGP_MC_URG=1" "MOGP_MC_WGS=2 MOGP_MC_PARK=1 MOGP_MC_URG=2"; do
    env $cfg B=$B N=2000 D=10 REPS=10 timeout 300 python tools/mchol_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[$cfg] /"
  done
done
for cfg in "MOGP_MC_WGS=1" "MOGP_MC_WGS=2 MOGP_MC_PARK=1" "MOGP_MC_WGS=2 MOGP_MC_PARK=0"; do
  env $cfg B=2 N=5000 D=20 REPS=6 timeout 300 python tools/mchol_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[$cfg] /"
  env $cfg B=4 N=5000 D=20 REPS=6 timeout 300 python tools/mchol_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[$cfg] /"
done
} 2>&1 | tee $O/sweep.txt
