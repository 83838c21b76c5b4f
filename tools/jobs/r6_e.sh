#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_e; rm -rf $O; mkdir -p $O; cd $R
{ BS=384,512 timeout 600 python tools/big_batch.py; BS=512,1024 N=1000 timeout 600 python tools/big_batch.py; BS=1024,2048 N=500 timeout 600 python tools/big_batch.py; BS=2048,4096 N=250 timeout 600 python tools/big_batch.py; BS=48,64 N=5000 D=20 timeout 600 python tools/big_batch.py; } > $O/big_batch.txt 2>&1; cat $O/big_batch.txt
