#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_i; rm -rf $O; mkdir -p $O; cd $R
rm -f /tmp/mc.trace
MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=8:2000:10 REPS=1 timeout 300 python tools/mchol_check.py > /dev/null 2>&1
python tools/mchol_trace.py /tmp/mc.trace -2 0 > $O/trace8.txt 2>&1
python tools/mchol_rows.py /tmp/mc.trace -2 0 9 > $O/rows8.txt 2>&1
head -30 $O/trace8.txt | cut -c1-200
