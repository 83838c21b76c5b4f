#!/bin/bash
# round 5: the barrier of mainloop_q pinned behind the step's MFMAs (PIN): loop probe, then in-tree (PIN) against .ab/lib_head.so (not pinned)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5n_${TAG:-head}; rm -rf $O; mkdir -p $O
{
timeout 300 tools/gemm_loop_probe.bin 64 40
timeout 300 tools/gemm_loop_probe.bin 64 40 | grep "per k-step"
for shp in ${SHAPES:-"64 2000 10" "16 2000 10" "8 2000 10" "1 2000 10" "16 5000 20" "1 16000 8"}; do
  set -- $shp
  for lib in .ab/lib_head.so intree .ab/lib_head.so intree; do
    p=/root/repo/$lib; [ "$lib" = intree ] && p=""
    echo "[$lib] $shp"
    MOGP_LIB_PATH=$p B=$1 N=$2 D=$3 M=${M:-256} REPS=${REPS:-8} timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|mchol"
  done
done
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/ab.txt
# raw per-task stamps of one 64 x n=2000 launch, for offline study of the tail (tools/mchol_trace.py reads it)
rm -f /tmp/mc.trace
MOGP_MC_TRACE=/tmp/mc.trace CONFIGS=64:2000:10 REPS=1 timeout 300 python tools/mchol_check.py > /dev/null 2>&1
python - <<'PY'
import numpy as np
raw = np.fromfile("/tmp/mc.trace", dtype=np.uint64)
# keep only the last launch
off, last = 0, 0
while off < raw.size:
    nb, nt, NP, g = [int(x) for x in raw[off:off + 4].astype(np.int64)]
    trw = g // 1000000
    last = off
    off += 4 + nb * nt * trw
raw[last:].tofile("/root/repo/gpurun_out/r5n_%s/mc64.trace" % __import__("os").environ.get("TAG", "head"))
PY
ls -la $O
