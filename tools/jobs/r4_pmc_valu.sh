#!/bin/bash
# round 4: vector-ALU issue of the HBM / VALU kernels (K build, gradient reduction, cross covariance) at 64 x n=2000 x d=10 (own pass, no trace domains)
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/pmc_valu; rm -rf $O; mkdir -p $O
cd /tmp
PMC_M=10000 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/sq -- python $R/tools/pmc_step.py > $O/run.log 2>&1
python $R/tools/pmc_summary.py $(find $O/sq -name "*.db") > $O/pmc_sq_valu_B64.txt 2>&1
rm -rf $O/sq
