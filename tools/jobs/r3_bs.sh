#!/bin/bash
# back-substitution chain: both tiles of the next chunk requested a step ahead -- against the start-of-session library
export TMPDIR=/tmp
cd /root/repo
H="MOGP_LIB_PATH=$PWD/build_ab/lib_head.so"
run() { echo "== $1"; env $1 WHAT=fit REPS=${2:-16} timeout 600 python tools/ab.py "$H" "" "$H" "" 2>&1 | tail -4 | cut -c1-150; }
run "B=64 N=2000 D=10 M=128" 12
run "B=8 N=2000 D=10 M=128"
run "B=16 N=5000 D=20 M=128 KERNEL=Matern52" 6
run "B=1 N=16000 D=8 M=128" 4
run "B=64 N=500 D=5 M=128"
for lib in build_ab/lib_head.so mogp_emulator_amd/libmogp_hip.so; do MOGP_LIB_PATH=$PWD/$lib B=8 REPS=8 M=256 timeout 300 python tools/kern_times.py 2>&1 | grep -E "fit|mchol|cov_build"; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backsol or timeout or schedules or c2_ or batch" 2>&1 | tail -2
