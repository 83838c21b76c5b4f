#!/bin/bash
# round 5: band-ahead task order (default) against the in-order table (MOGP_MC_AHEAD=0) by batch size; same bits; GPU suite
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r5s_${TAG:-head}; rm -rf $O; mkdir -p $O
{
MOGP_MC_AHEAD=0 timeout 900 python tools/factor_hash.py 2>&1 | grep CASE > $O/hash_inorder.txt
timeout 900 python tools/factor_hash.py 2>&1 | grep CASE > $O/hash_ahead.txt
if cmp -s $O/hash_inorder.txt $O/hash_ahead.txt; then echo "factor_hash: $(wc -l < $O/hash_ahead.txt) cases, band-ahead == in-order BIT FOR BIT"; else echo "factor_hash: DIFFERENT"; diff $O/hash_inorder.txt $O/hash_ahead.txt; fi
for shp in 1,2000,10 4,2000,10 8,2000,10 12,2000,10 16,2000,10 24,2000,10 32,2000,10 64,2000,10 2,5000,20 16,5000,20 1,16000,8 3,700,5; do
  set -- ${shp//,/ }
  for a in 0 1 0 1; do
    MOGP_MC_AHEAD=$a B=$1 N=$2 D=$3 REPS=10 timeout 300 python tools/mchol_time.py 2>&1 | grep -v amdgpu.ids | sed "s/checksum.*//"
  done
done
} 2>&1 | tee $O/ab.txt
if [ -n "$SUITE" ]; then
  timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > $O/gpu_tests.txt; tail -15 $O/gpu_tests.txt
fi
