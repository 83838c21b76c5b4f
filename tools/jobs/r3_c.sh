#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
for CFG in ${CFGS:-8:2000:10}; do
  rm -f /tmp/mc.trace
  MOGP_CHOL=mchol MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=$CFG REPS=1 timeout 300 python tools/mchol_check.py 2>&1 | tail -1 | cut -c1-120
  python tools/mchol_trace.py /tmp/mc.trace -2 0 > gpurun_out/r3c_trace_${CFG//:/_}.txt 2>&1; python - <<PY
import numpy as np
raw=np.fromfile("/tmp/mc.trace",dtype=np.uint64)
# keep only the second-to-last launch
off=0; L=[]
while off<raw.size:
    nb,nt,NP,g=[int(x) for x in raw[off:off+4].astype(np.int64)]; w=nb*nt*(g//1000000); L.append((off,4+w)); off+=4+w
o,n=L[-2]; raw[o:o+n].tofile("gpurun_out/r3c_raw_${CFG//:/_}.bin")
PY
  cat gpurun_out/r3c_trace_${CFG//:/_}.txt | cut -c1-260
done
if [ -n "$FULL" ]; then MOGP_MC_SPIN=400000 CONFIGS=${FULL} timeout 600 python tools/mchol_check.py 2>&1 | tail -14 | cut -c1-200; fi
