#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
export MOGP_MC_SPIN=300000 BASE_SCHED=5 REPS=5
echo "== split forced on, small first"
MOGP_MC_SPLIT=1 CONFIGS=3:700:5,2:300:3 timeout 120 python tools/mchol_check.py 2>&1 | tail -2 | cut -c1-230
MOGP_MC_SPLIT=1 CONFIGS=8:2000:10,16:2000:10,32:2000:10,64:2000:10,120:2000:10,16:5000:20:m,1:16000:8,2:5000:20:m timeout 600 python tools/mchol_check.py 2>&1 | tail -8 | cut -c1-230
echo "== split off"
MOGP_MC_SPLIT=0 CONFIGS=16:2000:10,32:2000:10,64:2000:10,120:2000:10,16:5000:20:m,1:16000:8 timeout 600 python tools/mchol_check.py 2>&1 | tail -6 | cut -c1-230
