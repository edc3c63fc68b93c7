#!/bin/bash
# SQ counter passes over the attention kernels (tools/attn_prof.py, interm_1b shape): where a wave's cycles go.
#   on the GPU box:  bash tools/attn_pmc.sh <tag> [lib.so]      -> gpurun_out/<tag>_attn_pmc.txt
TAG=${1:-attn}
LIB=${2:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$TAG
[ -n "$LIB" ] && export ORBIT2_HIP_LIB=$R/$LIB
rocprofv3 -L > $R/gpurun_out/pmc_$TAG/avail.txt 2>&1 || true
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_$TAG -o p$i -- python3 $R/tools/attn_prof.py > $R/gpurun_out/pmc_$TAG/p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
cd $R
python tools/summarize_prof.py sq gpurun_out/pmc_$TAG gpurun_out/${TAG}_attn_pmc.txt
cat gpurun_out/${TAG}_attn_pmc.txt
