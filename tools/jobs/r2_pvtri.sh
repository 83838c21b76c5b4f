#!/bin/bash
# SQ PMC pass + kernel trace of the predict kernels with / without the structural-zero skip (MOGP_PV_TRI)
export TMPDIR=/tmp
R=/root/repo
for t in 0 1; do
  cd /tmp
  rm -rf $R/gpurun_out/pvtri_sq$t $R/gpurun_out/pvtri_kt$t
  MOGP_PV_TRI=$t PMC_M=10000 timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/pvtri_sq$t -- python $R/tools/pmc_step.py > /dev/null 2>&1
  MOGP_PV_TRI=$t PMC_M=10000 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pvtri_kt$t -- python $R/tools/pmc_step.py > /dev/null 2>&1
  cd $R
  python tools/pmc_summary.py $(find gpurun_out/pvtri_sq$t -name "*.db") > gpurun_out/pvtri_sq$t.txt
  grep -E "^kernel|predict_var_w" gpurun_out/pvtri_sq$t.txt | cut -c1-200
  python tools/prof_summary.py $(find gpurun_out/pvtri_kt$t -name "*.db" | head -1) "pvtri $t" 2>&1 | grep -E "predict_var_w|Name|name" | cut -c1-200
done
