#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/k; export TMPDIR=/tmp
C=3:700:5,8:2000:10,16:2000:10,32:2000:10,64:2000:10,2:5000:20:m,16:5000:20:m,1:16000:8
rm -f gpurun_out/k/skew.txt
for S in 0 4 6 8 12 16 32; do
echo "== skew $S" >> gpurun_out/k/skew.txt
MOGP_MC_SKEW=$S BASE_SCHED=5 MOGP_MC_SPIN=400000 REPS=5 CONFIGS=$C timeout 900 python tools/mchol_check.py 2>&1 | tail -n 8 | cut -c1-108 >> gpurun_out/k/skew.txt
done
cat gpurun_out/k/skew.txt
