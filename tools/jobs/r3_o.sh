#!/bin/bash
# predictive variance: short row tile of a pair walked DOWN in k (MOGP_PV_DESC, default 1) against both passes upward
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3o; rm -rf $O; mkdir -p $O
WHAT=predict REPS=8 timeout 600 python tools/ab.py "MOGP_PV_DESC=0" "" "MOGP_PV_LGC=4" "MOGP_PV_LGC=2" > $O/ab.log 2>&1
tail -12 $O/ab.log
for d in 0 1; do MOGP_PV_DESC=$d REPS=4 timeout 300 python tools/kern_times.py 2>&1 | grep -E "predict|in-tree" > $O/kt_desc$d.log; cat $O/kt_desc$d.log; done
bash tools/pmc_fetch.sh $O/pmc_desc1.txt MOGP_PV_DESC=1 | grep -iE "predict_var|kernel" 
bash tools/pmc_fetch.sh $O/pmc_desc0.txt MOGP_PV_DESC=0 | grep -iE "predict_var|kernel"
grep -i predict_var $O/pmc_desc1.txt $O/pmc_desc0.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullbatch.py -m gpu -x -q -k "predict or c3_full" 2>&1 | tail -3
