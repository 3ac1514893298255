#!/bin/bash
mkdir -p gpurun_out/pmc_kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_kernels/set$i -o p -- python3 $R/tools/pmc_kernels.py > $R/gpurun_out/pmc_kernels/set$i.log 2>&1
  tail -1 $R/gpurun_out/pmc_kernels/set$i.log | cut -c1-200
done
cd $R
python3 tools/pmc_kernels_summary.py gpurun_out/pmc_kernels gpurun_out/pmc_kernels_summary.json | cut -c1-200 | head -150
