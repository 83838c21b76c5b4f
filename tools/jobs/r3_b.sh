#!/bin/bash
# round 3, run B: first run of the one-launch Cholesky
export TMPDIR=/tmp
cd /root/repo
MOGP_MC_SPIN=200000 CONFIGS=1:100:3 REPS=2 timeout 120 python tools/mchol_check.py 2>&1 | tail -5
MOGP_MC_SPIN=200000 CONFIGS=5:130:4,3:700:5 REPS=2 timeout 120 python tools/mchol_check.py 2>&1 | tail -5
MOGP_MC_SPIN=400000 timeout 600 python tools/mchol_check.py 2>&1 | tail -14
