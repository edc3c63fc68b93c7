for w in 8 0; do for pin in 1 0; do
ORBIT2_NPZ_PIN=$pin python bench.py --data npz --data-workers $w --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);n=d['data_npz'];print('workers $w pin $pin: resident %.2f  npz %.2f samples/s  %.0f ms/step  wait %.0f ms  host-in-step %.0f ms  loader cpu %.3f s/sample'%(d['value'],n['value'],n['ms_per_step'],n['consumer_wait_ms_per_step'],n['host_ms_per_step_in_the_step_call'],n['loader_cpu_s_per_sample']))"
done; done
