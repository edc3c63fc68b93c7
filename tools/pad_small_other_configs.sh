for v in 0 1 0 1; do
ORBIT2_LD_PAD_SMALL=$v python bench.py --model interm_117m --grid 32x64 --batch 8 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('117m pad_small=$v: %.1f samples/s %.3f ms/step'%(d['value'],d['ms_per_step']))"
done
for v in 0 1; do
ORBIT2_LD_PAD_SMALL=$v python bench.py --daymet --grid 96x192 --batch 4 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('daymet pad_small=$v: %.2f samples/s %.2f ms/step'%(d['value'],d['ms_per_step']))"
done
