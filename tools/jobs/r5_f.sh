#!/bin/bash
# round 5: LDS bank conflicts / LDS activity of the GEMM main-loop probe variants (own PMC pass, no trace domains).  usage: TAG=x bash tools/jobs/r5_f.sh
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r5f_${TAG:-head}; rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc -- $R/tools/gemm_loop_probe.bin 64 10 > $O/run.log 2>&1
python $R/tools/pmc_summary.py $(find $O/pmc -name "*.db") > $O/pmc_lds.txt 2>&1
rm -rf $O/pmc
cat $O/pmc_lds.txt | cut -c1-260
tail -14 $O/run.log
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "c5_like_conditioning or c5_full_size or grouped_columns or sharded_wrapper or ragged" 2>&1 | tail -25 | tee $O/new_tests.txt
