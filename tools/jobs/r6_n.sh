#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_n; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python tools/fitmap_context.py 2>&1 | grep -v amdgpu > $O/out.txt; cat $O/out.txt
