#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
rm -f /tmp/mc.trace
MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=${CFG:-8:2000:10} REPS=1 timeout 300 python tools/mchol_check.py 2>&1 | tail -3
python tools/mchol_trace.py /tmp/mc.trace -2 0 > gpurun_out/r3c_trace.txt 2>&1
cat gpurun_out/r3c_trace.txt | cut -c1-520
if [ -n "$FULL" ]; then MOGP_MC_SPIN=400000 CONFIGS=${FULL} timeout 600 python tools/mchol_check.py 2>&1 | tail -14; fi
