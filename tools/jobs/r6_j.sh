#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_j; rm -rf $O; mkdir -p $O; cd $R
{ B=64 timeout 600 python tools/fitmap_timing.py; B=64 PROF=1 timeout 600 python tools/fitmap_timing.py;  B=8 timeout 600 python tools/fitmap_timing.py; for r in 128 192 320 384; do echo "MOGP_START_REPLICAS=$r"; MOGP_START_REPLICAS=$r B=64 timeout 600 python tools/fitmap_timing.py | tail -2; done; } 2>&1 | grep -v amdgpu > $O/fitmap.txt; cat $O/fitmap.txt
