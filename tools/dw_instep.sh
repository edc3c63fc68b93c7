#!/bin/bash
# Round 6, verdict item 2, in the STEP: the default bench step with the weight-gradient loop (TN form only) under the K-wrap ablation
# (activations stay right; only the weight gradients are wrong) -> per-launch time of gemm256w_grouped_kernel<2> per arm.
R=${GRAFT_REPO_ROOT:-/root/repo}
ALT=$R/orbit-2_amd/lib/alt
OUT=$R/gpurun_out/${TAG:-r06_dw_traffic_instep}.txt
mkdir -p $R/gpurun_out/dw_instep
cd /tmp && export TMPDIR=/tmp
echo "# bench.py --steps 4 --warmup 2 under rocprofv3 --kernel-trace --stats; arms: base, kwraptn64 (panels from the Infinity Cache), kwraptn2 (from L2)" > $OUT
for arm in base ${ARMS:-kwraptn64 kwraptn2}; do
  lib=$ALT/$arm.so; [ $arm = base ] && lib=$R/orbit-2_amd/lib/liborbit2_hip.so
  export ORBIT2_HIP_LIB=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dw_instep -o $arm -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/dw_instep/$arm.json 2> $R/gpurun_out/dw_instep/$arm.err || echo "arm $arm failed" >> $OUT
  f=$(find $R/gpurun_out/dw_instep -name "${arm}_kernel_stats.csv" | head -1)
  echo "== $arm: $(python3 -c "import json,sys; d=json.loads(open('$R/gpurun_out/dw_instep/$arm.json').read().strip().splitlines()[-1]); print('%.3f samples/s %.2f ms/step' % (d['value'], d['ms_per_step']))" 2>&1)" >> $OUT
  grep -E "gemm256w" $f | sed -e "s/void (anonymous namespace):://" -e "s/(.*)\"/\"/" | cut -d, -f1-5 >> $OUT
done
cat $OUT
