#!/bin/bash
# round 5: in-tree library against .ab/lib_head.so: same bits (tools/factor_hash.py), then mchol / fit times at several batch sizes
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5q_${TAG:-head}; rm -rf $O; mkdir -p $O
{
MOGP_LIB_PATH=/root/repo/.ab/lib_head.so timeout 900 python tools/factor_hash.py 2>&1 | grep CASE > $O/hash_head.txt
timeout 900 python tools/factor_hash.py 2>&1 | grep CASE > $O/hash_intree.txt
if cmp -s $O/hash_head.txt $O/hash_intree.txt; then echo "factor_hash: $(wc -l < $O/hash_intree.txt) cases, in-tree == lib_head BIT FOR BIT"; else echo "factor_hash: DIFFERENT"; diff $O/hash_head.txt $O/hash_intree.txt; fi
for shp in ${SHAPES:-64,2000,10 32,2000,10 16,2000,10 8,2000,10 1,2000,10 16,5000,20 1,16000,8}; do
  set -- ${shp//,/ }
  for lib in .ab/lib_head.so intree .ab/lib_head.so intree; do
    p=/root/repo/$lib; [ "$lib" = intree ] && p=""
    echo "[$lib] $shp"
    MOGP_LIB_PATH=$p B=$1 N=$2 D=$3 M=${M:-256} REPS=${REPS:-8} timeout 600 python tools/kern_times.py 2>&1 | grep "fit \|mchol"
  done
done
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/ab.txt
if [ -n "$SUITE" ]; then
  timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > $O/gpu_tests.txt; tail -15 $O/gpu_tests.txt
fi
