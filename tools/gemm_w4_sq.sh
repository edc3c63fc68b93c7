#!/bin/bash
# SQ counter passes over the two 256-tile GEMM kernels (tools/gemm_w4_pmc.py: qkv and fc2 forward shapes, hints 256 / 260):
# where a wave's cycles go.   on the GPU box:  bash tools/gemm_w4_sq.sh  -> gpurun_out/r03_gemm_w4_sq.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_w4sq
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_w4sq -o p$i -- python3 $R/tools/gemm_w4_pmc.py > $R/gpurun_out/pmc_w4sq/p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
cd $R
python tools/summarize_prof.py sq gpurun_out/pmc_w4sq gpurun_out/r03_gemm_w4_sq.txt
cat gpurun_out/r03_gemm_w4_sq.txt
