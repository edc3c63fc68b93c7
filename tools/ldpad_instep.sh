#!/bin/bash
# Round 6: the step with _ld_pad's rule widened (rows a multiple of MOD bytes apart get 64 elements of padding): MOD = 8192 (rounds 3-5) vs 2048
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_ld_pad_instep.txt
mkdir -p $R/gpurun_out/ldpad
cd /tmp && export TMPDIR=/tmp
echo "# bench.py --steps 4 --warmup 2 under rocprofv3 --kernel-trace --stats; ORBIT2_LD_PAD_MOD = 8192 | 2048 | 8192 | 2048" > $OUT
i=0
for mod in 8192 2048 8192 2048; do
  i=$((i+1))
  export ORBIT2_LD_PAD_MOD=$mod
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ldpad -o m${mod}_$i -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > $R/gpurun_out/ldpad/m${mod}_$i.json 2> $R/gpurun_out/ldpad/m${mod}_$i.err || echo "run $mod failed" >> $OUT
  f=$(find $R/gpurun_out/ldpad -name "m${mod}_${i}_kernel_stats.csv" | head -1)
  echo "== MOD $mod: $(python3 -c "import json,sys; d=json.loads(open('$R/gpurun_out/ldpad/m${mod}_$i.json').read().strip().splitlines()[-1]); print('%.3f samples/s %.2f ms/step loss %.5f' % (d['value'], d['ms_per_step'], d['step_model']['final_loss']))" 2>&1)" >> $OUT
  grep -E "gemm256w|attn_|ln_|dropout_bwd|colsum_part" $f | sed -e "s/void (anonymous namespace):://" -e "s/(.*)\"/\"/" | cut -d, -f1-4 >> $OUT
done
cat $OUT
