#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_m; rm -rf $O; mkdir -p $O; cd $R
{ B=64 NUGGET=1e-6 timeout 600 python tools/fitmap_timing.py; B=64 NUGGET=fit timeout 600 python tools/fitmap_timing.py; B=64 NUGGET=1e-6 PROF=1 timeout 600 python tools/fitmap_timing.py;
  for g in 0 1; do MOGP_GRAD_CHAIN=$g WHAT=fit,grad B=64 timeout 300 python tools/ab.py ""; MOGP_GRAD_CHAIN=$g WHAT=fit,grad B=8 timeout 300 python tools/ab.py ""; MOGP_GRAD_CHAIN=$g WHAT=fit,grad B=1 timeout 300 python tools/ab.py ""; done; } 2>&1 | grep -v amdgpu > $O/out.txt; cat $O/out.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or medium or fixture or grid11 or c2_full or switches or backsolve or non_finite or multioutput or tile_boundary" 2>&1 | tail -4
