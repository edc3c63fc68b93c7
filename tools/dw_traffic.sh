#!/bin/bash
# Round 6, verdict item 2 (on the GPU box):  bash tools/dw_traffic.sh  -> gpurun_out/r06_dw_traffic_clock.txt
# arms: the product library and the kwrap ablations (built here by tools/mkvar_w4.sh before the push, or on the box)
R=${GRAFT_REPO_ROOT:-/root/repo}
ALT=$R/orbit-2_amd/lib/alt
OUT=$R/gpurun_out/${TAG:-r06_dw_traffic_clock}.txt
mkdir -p $R/gpurun_out/pmc_dw
cd /tmp && export TMPDIR=/tmp
{
echo "# grouped dW (gemm256w_grouped_kernel<2>): the Block's four weight-gradient products at batch 16 (131072 tokens)"
echo "# arms: base = the product loop; kwrapN = TIMING-ONLY ablation, the K walk wraps over the first N K-tiles (panels from L2)"
echo "## interleaved timing, one process"
ARMS=${ARMS:-"kwrap8 kwrap2"}
libs=""; for a in $ARMS; do libs="$libs $ALT/$a.so"; done
python3 $R/tools/dw_traffic_ab.py $R/orbit-2_amd/lib/liborbit2_hip.so $libs 2>&1 | grep -v -e Warning -e amdgpu.ids
echo "## counters, one rocprofv3 pass per arm and counter set (14 launches, the last 8 summarised)"
} > $OUT
for arm in base $ARMS; do
  lib=$ALT/$arm.so; [ $arm = base ] && lib=$R/orbit-2_amd/lib/liborbit2_hip.so
  export ORBIT2_HIP_LIB=$lib
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_dw -o ${arm}_mfma -- python3 $R/tools/dw_traffic_ab.py --pmc > $R/gpurun_out/pmc_dw/${arm}_mfma.log 2>&1 || echo "pass ${arm}_mfma failed" >> $OUT
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_dw -o ${arm}_fetch -- python3 $R/tools/dw_traffic_ab.py --pmc > $R/gpurun_out/pmc_dw/${arm}_fetch.log 2>&1 || echo "pass ${arm}_fetch failed" >> $OUT
  echo "arm $arm done"
done
unset ORBIT2_HIP_LIB
cd $R
python3 tools/dw_traffic_sum.py gpurun_out/pmc_dw base $ARMS >> $OUT
cat $OUT
