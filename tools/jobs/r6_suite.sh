#!/bin/bash
export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6_suite; rm -rf $O; mkdir -p $O; cd $R
cp mogp_emulator_amd/libmogp_hip.build $O/build_commit.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=10 2>&1 | tail -40 > $O/gpu_tests.txt
tail -30 $O/gpu_tests.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/gpu_tests.txt
