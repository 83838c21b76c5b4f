#!/bin/bash
# raw per-task traces of the 8 x n=2000 one-launch Cholesky for the task orders 1 and 3 (analysed on the host: tools/mchol_band.py)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r3tr; rm -rf $O; mkdir -p $O
for la in 1 3; do rm -f /tmp/mc.trace; MOGP_MC_LA=$la MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=8:2000:10 REPS=1 timeout 300 python tools/mchol_check.py > /dev/null 2>&1
  python - <<PY
import numpy as np
raw=np.fromfile('/tmp/mc.trace',dtype=np.uint64)
# keep the last-but-one launch only
off=0; L=[]
while off<raw.size:
    nb,nt,NP,g=[int(x) for x in raw[off:off+4].astype(np.int64)]; trw=g//1000000 if g>=1000000 else 8
    w=nb*nt*trw; L.append((off,4+w)); off+=4+w
o,n=L[-2]; raw[o:o+n].tofile('$O/trace_la$la.bin')
PY
done
ls -la $O
