#!/bin/bash
# L2-miss traffic of the hot kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_step.py,
# summarised per kernel by tools/pmc_summary.py.  Usage (on the GPU box): bash tools/pmc_fetch.sh <out-file> [ENV=...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/pmc_f /root/repo/gpurun_out/pmc_w
env PMC_M=10000 "$@" timeout 300 rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/pmc_f -- python /root/repo/tools/pmc_step.py > /dev/null 2>&1
env PMC_M=10000 "$@" timeout 300 rocprofv3 --pmc WRITE_SIZE -d /root/repo/gpurun_out/pmc_w -- python /root/repo/tools/pmc_step.py > /dev/null 2>&1
cd /root/repo
python tools/pmc_summary.py $(find gpurun_out/pmc_f gpurun_out/pmc_w -name "*.db") > "$out"
head -6 "$out"
