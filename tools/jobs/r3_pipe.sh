#!/bin/bash
# mainloop_w: LDS fragment reads of k-slice kk+1 in front of the MFMAs of slice kk, order kept with sched_group_barrier -- vs the previous build
# (the pipelined fragment reads were an experiment of this job only: not in the tree, DESIGN.md section 5 list)
export TMPDIR=/tmp
cd /root/repo
P="MOGP_LIB_PATH=$PWD/build_ab/lib_prev.so"
WHAT=predict REPS=8 timeout 900 python tools/ab.py "$P" "" "$P" "" 2>&1 | tail -4 | cut -c1-160
for lib in build_ab/lib_prev.so mogp_emulator_amd/libmogp_hip.so; do MOGP_LIB_PATH=$PWD/$lib REPS=4 timeout 300 python tools/kern_times.py 2>&1 | grep -E "predict_var"; done
B=16 N=5000 D=20 M=10000 KERNEL=Matern52 WHAT=predict REPS=4 timeout 900 python tools/ab.py "$P" "" 2>&1 | tail -2 | cut -c1-160
B=1 N=16000 D=8 M=10000 WHAT=fit,predict REPS=4 timeout 900 python tools/ab.py "$P" "" 2>&1 | tail -2 | cut -c1-200
