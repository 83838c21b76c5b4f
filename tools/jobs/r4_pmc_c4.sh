#!/bin/bash
# round 4: SQ counters of the C4 workload (16 x n=5000 x d=20, Matern-5/2): is its K build vector-ALU-bound?  (own pass, no trace domains)
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/pmc_c4; rm -rf $O; mkdir -p $O
cd /tmp
PMC_B=16 PMC_N=5000 PMC_D=20 PMC_M=2048 PMC_KERNEL=Matern52 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/sq -- python $R/tools/pmc_step.py > $O/run.log 2>&1
python $R/tools/pmc_summary.py $(find $O/sq -name "*.db") > $O/pmc_sq_C4.txt 2>&1
rm -rf $O/sq
head -30 $O/pmc_sq_C4.txt | cut -c1-250
