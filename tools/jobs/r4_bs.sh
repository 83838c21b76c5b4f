#!/bin/bash
# round 4: back substitution chain whose chunks poll the solution VALUES (preset to an all-ones pattern by the K build; MOGP_BS_SENTINEL=1)
# against the flag + payload hand-off (=0)
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/bs_${TAG:-head}; mkdir -p $O
{
for shp in ${SHAPES:-1:2000:10 8:2000:10 16:2000:10 64:2000:10 2:5000:20 1:16000:8 3:700:5 1:100:3}; do
  IFS=':' read -r b n d <<< "$shp"
  echo "== $shp"
  B=$b N=$n D=$d WHAT=fit,predict M=500 REPS=${REPS:-15} timeout 600 python tools/ab.py "MOGP_BS_SENTINEL=0" "MOGP_BS_SENTINEL=1" "MOGP_BS_SENTINEL=0" "MOGP_BS_SENTINEL=1"
done
} 2>&1 | grep -v "^$" | tee $O/ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backsolve or back_sub or hand_offs or fullsize or C2 or batch or timeout or switches" 2>&1 | tail -5 > $O/tests.txt
