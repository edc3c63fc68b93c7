#!/bin/bash
# Same-box A/B of the default bench step under an environment switch of the library / the Python layer:
#   bash tools/env_instep.sh TAG VAR value1 value2 [value1 value2 ...]   -> gpurun_out/TAG.txt (per-kernel ms/step of the largest kernels per run)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; VAR=$2; shift 2
OUT=$R/gpurun_out/$TAG.txt
mkdir -p $R/gpurun_out/envab
cd /tmp && export TMPDIR=/tmp
echo "# bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs under rocprofv3 --kernel-trace --stats; $VAR = $*" > $OUT
i=0
for v in "$@"; do
  i=$((i+1))
  export $VAR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/envab -o ${TAG}_$i -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > $R/gpurun_out/envab/${TAG}_$i.json 2> $R/gpurun_out/envab/${TAG}_$i.err || echo "run $v failed" >> $OUT
  echo "== $VAR=$v: $(python3 -c "import json,sys; d=json.loads(open('$R/gpurun_out/envab/${TAG}_$i.json').read().strip().splitlines()[-1]); print('%.3f samples/s %.2f ms/step loss %.5f' % (d['value'], d['ms_per_step'], d['step_model']['final_loss']))" 2>&1)" >> $OUT
done
cd $R
python3 - "$TAG" "$@" >> $OUT <<'PY'
import csv, glob, sys
tag, vals = sys.argv[1], sys.argv[2:]
res = []
for i, v in enumerate(vals, 1):
    f = glob.glob("gpurun_out/envab/**/%s_%d_kernel_stats.csv" % (tag, i), recursive=True)
    d = {}
    if f:
        for r in csv.DictReader(open(f[0])):
            d[r["Name"].replace("void (anonymous namespace)::", "").split("(")[0][:44]] = float(r["TotalDurationNs"]) / 1e6 / 6
    res.append(d)
keys = sorted(res[0], key=lambda k: -res[0][k])[:22]
print("%-46s" % "kernel, ms per step" + " ".join("%10s" % v for v in vals))
for k in keys:
    print("%-46s" % k + " ".join("%10.2f" % d.get(k, 0) for d in res))
print("%-46s" % "ALL KERNELS" + " ".join("%10.2f" % sum(d.values()) for d in res))
PY
cat $OUT
