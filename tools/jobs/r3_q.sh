#!/bin/bash
# predictive variance: persistent workgroups + soft lock-step (MOGP_PV_SYNC polls, MOGP_PV_LAG blocks of slack)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3q; rm -rf $O; mkdir -p $O
WHAT=predict REPS=8 timeout 900 python tools/ab.py "MOGP_PV_SYNC=0" "" "MOGP_PV_SYNC=1000" "MOGP_PV_SYNC=1000 MOGP_PV_LAG=1" "MOGP_PV_SYNC=1000 MOGP_PV_LAG=2" "MOGP_PV_SYNC=1 MOGP_PV_LAG=20" "MOGP_PV_SYNC=1000 MOGP_PV_DESC=0" > $O/ab.log 2>&1
tail -9 $O/ab.log
for cfg in "MOGP_PV_SYNC=1000" "MOGP_PV_SYNC=1000 MOGP_PV_LAG=1" "MOGP_PV_SYNC=1000 MOGP_PV_LAG=2" "MOGP_PV_SYNC=1 MOGP_PV_LAG=20"; do
  echo "== $cfg"
  bash tools/pmc_fetch.sh $O/pmc.txt $cfg | grep -iE "predict_var"
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullbatch.py -m gpu -x -q -k "predict or c3_full" 2>&1 | tail -3
