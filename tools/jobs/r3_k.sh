#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
for l in pd2 pd4 pd6; do
  for cfg in "B=64 N=2000 D=10 M=256" "B=16 N=5000 D=20 M=256" "B=1 N=16000 D=8 M=256" "B=8 N=2000 D=10 M=256"; do
    env $cfg REPS=5 MOGP_LIB_PATH=$PWD/build_ab/lib_$l.so python tools/kern_times.py 2>&1 | grep -E "fit |mchol" | tr '\n' ' '; echo " [$l $cfg]"
  done
done
MOGP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c1-600
