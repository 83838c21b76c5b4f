#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_q; rm -rf $O; mkdir -p $O; cd $R
FUZZ_ONLY=530,1360 timeout 600 python tests/tools/fuzz_parity.py 1500 612 2>&1 | tail -6 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullbatch.py -m gpu -q -x -k "golden or medium or fixture or grid11 or c2_full or c3_full or c4 or gradient or objective_with or fit_GP_MAP or matern or tile_boundary or multioutput or mean" 2>&1 | tail -6 | cut -c1-200
{ WHAT=fit,grad B=64 REPS=10 timeout 300 python tools/ab.py ""; WHAT=fit,grad B=64 REPS=10 timeout 300 python tools/ab.py ""; WHAT=fit,grad B=16 N=5000 D=20 KERNEL=Matern52 REPS=5 timeout 300 python tools/ab.py ""; WHAT=fit,grad B=32 REPS=10 timeout 300 python tools/ab.py ""; B=64 timeout 300 python tools/fitmap_timing.py | tail -2; } 2>&1 | grep -v amdgpu > $O/out.txt; cat $O/out.txt
