#!/bin/bash
# round 5: predictive variance on the q loop (predict_var_q_kernel, default) against the 8-wave 128 x 128 kernel (MOGP_PV_Q=0)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5pv_${TAG:-head}; rm -rf $O; mkdir -p $O
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullbatch.py -m gpu -q -x -k "predict or variance or c2 or c3 or ragged" 2>&1 | tail -3
for shp in 64,2000,10 8,2000,10 1,2000,10 16,5000,20 1,16000,8; do
  set -- ${shp//,/ }
  for cfg in "MOGP_PV_Q=0" "MOGP_PV_Q=1" "MOGP_PV_Q=1 MOGP_PV_LGC=4" "MOGP_PV_Q=1 MOGP_PV_LGC=2" "MOGP_PV_Q=0" "MOGP_PV_Q=1"; do
    echo "[$cfg] $shp"
    env $cfg B=$1 N=$2 D=$3 M=10000 REPS=4 timeout 600 python tools/kern_times.py 2>&1 | grep "predict_var\|cross_cov"
  done
done
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/ab.txt
