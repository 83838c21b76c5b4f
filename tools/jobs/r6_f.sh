#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_f; rm -rf $O; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_reference_gpu_cases.py -m gpu -q -x -k "fit_GP_MAP or fitmap or tsunami or MAP" 2>&1 | tail -12 > $O/tests.txt; cat $O/tests.txt | cut -c1-200
{ for b in 64 16 8; do B=$b timeout 600 python tools/fitmap_timing.py; done; B=64 MAXITER=100 timeout 600 python tools/fitmap_timing.py; B=16 N=5000 D=20 TRIES=5 timeout 600 python tools/fitmap_timing.py; } > $O/fitmap.txt 2>&1; cat $O/fitmap.txt
