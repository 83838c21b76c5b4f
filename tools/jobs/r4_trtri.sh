#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/trtri_${TAG:-head}; mkdir -p $O
{
WHAT=grad REPS=8 timeout 900 python tools/ab.py "MOGP_LIB_PATH=/root/repo/$REF" "MOGP_TRTRI_WT4_FROM=100000" "MOGP_TRTRI_WT4_FROM=1024" "MOGP_TRTRI_WT4_FROM=512" "MOGP_TRTRI_WT4_FROM=256" "MOGP_TRTRI_WT4_FROM=128"
B=16 N=5000 D=20 WHAT=grad REPS=4 timeout 900 python tools/ab.py "MOGP_LIB_PATH=/root/repo/$REF" "MOGP_TRTRI_WT4_FROM=100000" "MOGP_TRTRI_WT4_FROM=1024" "MOGP_TRTRI_WT4_FROM=512" "MOGP_TRTRI_WT4_FROM=256"
B=8 WHAT=grad REPS=8 timeout 900 python tools/ab.py "MOGP_LIB_PATH=/root/repo/$REF" "MOGP_TRTRI_WT4_FROM=100000" "MOGP_TRTRI_WT4_FROM=512" "MOGP_TRTRI_WT4_FROM=256"
B=1 N=16000 D=8 WHAT=grad REPS=3 timeout 900 python tools/ab.py "MOGP_LIB_PATH=/root/repo/$REF" "MOGP_TRTRI_WT4_FROM=100000" "MOGP_TRTRI_WT4_FROM=512" "MOGP_TRTRI_WT4_FROM=256"
} 2>&1 | grep -v "^$" | tee $O/trtri.txt
