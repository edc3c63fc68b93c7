for b in 16 24 32 16 32; do
python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('batch $b: %.3f samples/s %.1f ms/step frac %.4f'%(d['value'],d['ms_per_step'],d['roofline']['frac']))"
done
