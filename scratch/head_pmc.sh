#!/bin/bash
# SQ counters of head16_kernel / head_reg_kernel (separate passes; --kernel-trace only beside --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02c
for abl in 0 3; do
  rm -rf /tmp/pmcA /tmp/pmcB
  DCLR_HR_ABL=$abl rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d /tmp/pmcA -- python3 $R/scratch/head_pmc.py 8192 > /dev/null 2>&1
  DCLR_HR_ABL=$abl rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d /tmp/pmcB -- python3 $R/scratch/head_pmc.py 8192 > /dev/null 2>&1
  echo "===== ABL=$abl"; python3 $R/scratch/pmc_summary.py /tmp/pmcA head; python3 $R/scratch/pmc_summary.py /tmp/pmcB head
done
