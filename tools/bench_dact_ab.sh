#!/bin/bash
# same-box A/B of the GELU-backward form (climate_learn/_ops.py:_DACT): bench.py with the factor stored by the forward (1) and with
# round 2's form (0: pre-activation stored, GELU' + mask in the backward epilogue), alternating
cd $GRAFT_REPO_ROOT
for r in 1 2; do for d in 1 0; do
python -c "
import sys, runpy
sys.path[:0] = ['$GRAFT_REPO_ROOT', '$GRAFT_REPO_ROOT/orbit-2_amd']
from climate_learn import _ops
_ops._DACT = bool($d)
sys.argv = ['bench.py', '--steps', '6', '--warmup', '2', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')
" > gpurun_out/dact_ab_$d.json 2>/dev/null
python -c "
import json
d = json.loads(open('gpurun_out/dact_ab_$d.json').read().strip().splitlines()[-1])
print('_DACT=$d  %.3f samples/s  %.1f ms/step  gemm roofline %.4f  loss %.5f' % (d['value'], d['ms_per_step'], d['roofline_gemm']['frac'], d['step_model']['final_loss']), flush=True)"
done; done
