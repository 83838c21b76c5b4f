#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/k; export TMPDIR=/tmp
C=3:700:5,8:2000:10,16:2000:10,32:2000:10,64:2000:10,2:5000:20:m,16:5000:20:m,1:16000:8
echo "== new" > gpurun_out/k/pipe.txt
BASE_SCHED=5 MOGP_MC_SPIN=400000 CONFIGS=$C timeout 900 python tools/mchol_check.py 2>&1 | tail -n 9 | cut -c1-230 >> gpurun_out/k/pipe.txt
echo "== old" >> gpurun_out/k/pipe.txt
MOGP_LIB_PATH=/root/repo/build_ab/lib_pre_kinv8.so BASE_SCHED=5 MOGP_MC_SPIN=400000 CONFIGS=$C timeout 900 python tools/mchol_check.py 2>&1 | tail -n 9 | cut -c1-230 >> gpurun_out/k/pipe.txt
cat gpurun_out/k/pipe.txt
CFGS="8:2000:10" bash tools/jobs/r3_c.sh 2>&1 | head -30 | cut -c1-230
