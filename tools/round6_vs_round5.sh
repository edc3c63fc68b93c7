#!/bin/bash
# Same-box A/B of everything round 6 changed on the headline step: arm r05 = the old tile walk (lib/alt/walk0.so: -DO2_W4_WALK=0
# -DO2_W4_TWALK=0, which also switches the cohort start barrier off), no balanced weight-gradient launch, round-5 row pitches;
# arm r06 = the defaults.  bench.py --steps 6 --warmup 2, alternating, no profiler.   -> gpurun_out/r06_vs_r05_same_box.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_vs_r05_same_box.txt
echo "# bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs, alternating on one box" > $OUT
for i in 1 2 3; do
  ORBIT2_HIP_LIB=$R/orbit-2_amd/lib/alt/walk0.so ORBIT2_DW_BALANCE=0 ORBIT2_LD_PAD_SMALL=0 ORBIT2_W4_PACE=0 python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('r05-equivalent: %.3f samples/s %.2f ms/step roofline.frac %.4f gemm %.4f'%(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline_gemm']['frac']))" >> $OUT
  python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('round 6       : %.3f samples/s %.2f ms/step roofline.frac %.4f gemm %.4f'%(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline_gemm']['frac']))" >> $OUT
done
cat $OUT
