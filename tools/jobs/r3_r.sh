#!/bin/bash
# predictive variance: why is the persistent form slower?  static / dynamic block numbers, hooks, vs the HEAD library
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3r; rm -rf $O; mkdir -p $O
WHAT=predict REPS=8 timeout 900 python tools/ab.py "MOGP_LIB_PATH=$PWD/build_ab/lib_head.so" "MOGP_PV_SYNC=0" "MOGP_PV_SYNC=0 MOGP_PV_PERSIST=1" "MOGP_PV_SYNC=0 MOGP_PV_PERSIST=2" "MOGP_PV_SYNC=1000" "MOGP_PV_SYNC=0 MOGP_PV_PERSIST=1 MOGP_PV_GRID=256"  "MOGP_LIB_PATH=$PWD/build_ab/lib_head.so" "MOGP_PV_SYNC=0" > $O/ab.log 2>&1
tail -9 $O/ab.log
