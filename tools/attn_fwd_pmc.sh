#!/bin/bash
# SQ counter passes over the attention FORWARD alone (tools/attn_fwd_prof.py): where a wave's cycles go, per library variant.
#   on the GPU box:  bash tools/attn_fwd_pmc.sh <tag> <B> <p> <flags> [lib.so ...]   -> gpurun_out/<tag>_<lib>_fwd_pmc.txt
TAG=$1; B=$2; P=$3; FL=$4; shift 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for LIB in "${@:-default}"; do
  N=$(basename $LIB .so)
  D=$R/gpurun_out/pmc_${TAG}_$N
  mkdir -p $D
  if [ "$LIB" != default ]; then export ORBIT2_HIP_LIB=$R/$LIB; else unset ORBIT2_HIP_LIB; fi
  i=0
  for set in \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
    "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
    "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE" ; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -o p$i -- python3 $R/tools/${PROF:-attn_fwd_prof.py} $B $P $FL > $D/p$i.log 2>&1 || echo "pass $i failed"
  done
  (cd $R && python tools/summarize_prof.py sq gpurun_out/pmc_${TAG}_$N gpurun_out/${TAG}_${N}_fwd_pmc.txt && cat gpurun_out/${TAG}_${N}_fwd_pmc.txt)
done
