#!/bin/bash
# predictive variance: soft lock-step of a super-tile's workgroups (MOGP_PV_SYNC polls, 0 = free-running) x descending short pass
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3p; rm -rf $O; mkdir -p $O
WHAT=predict REPS=8 timeout 900 python tools/ab.py "MOGP_PV_SYNC=0" "" "MOGP_PV_SYNC=8" "MOGP_PV_SYNC=1000" "MOGP_PV_DESC=0" "MOGP_PV_LGC=4" "MOGP_PV_LGC=2" > $O/ab.log 2>&1
tail -9 $O/ab.log
for sy in 64 0; do MOGP_PV_SYNC=$sy REPS=4 timeout 300 python tools/kern_times.py 2>&1 | grep -E "predict|in-tree" > $O/kt_sync$sy.log; cat $O/kt_sync$sy.log; done
bash tools/pmc_fetch.sh $O/pmc_sync64.txt MOGP_PV_SYNC=64 | grep -iE "predict_var|kernel" 
bash tools/pmc_fetch.sh $O/pmc_sync64_desc0.txt MOGP_PV_SYNC=64 MOGP_PV_DESC=0 | grep -iE "predict_var|kernel" 
bash tools/pmc_fetch.sh $O/pmc_sync64_lgc4.txt MOGP_PV_SYNC=64 MOGP_PV_LGC=4 | grep -iE "predict_var|kernel" 
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullbatch.py -m gpu -x -q -k "predict or c3_full" 2>&1 | tail -3
