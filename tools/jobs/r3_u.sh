#!/bin/bash
# 32-bit offset addressing in the other two MFMA main loops (gemm_mainloop, mainloop_pf): against the HEAD library, bit-identical?
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3u; rm -rf $O; mkdir -p $O
H="MOGP_LIB_PATH=$PWD/build_ab/lib_head.so"
WHAT=fit,grad REPS=12 timeout 600 python tools/ab.py "$H" "" "$H" "" 2>&1 | tail -4
B=8 WHAT=fit,grad REPS=12 timeout 600 python tools/ab.py "$H" "" "$H" "" 2>&1 | tail -4
B=16 N=5000 D=20 M=1000 KERNEL=Matern52 WHAT=fit,grad REPS=6 timeout 600 python tools/ab.py "$H" "" 2>&1 | tail -2
B=1 N=16000 D=8 M=1000 WHAT=fit,grad REPS=4 timeout 600 python tools/ab.py "$H" "" 2>&1 | tail -2
