# two data-parallel ranks on ONE card over gloo (rehearsal of the multi-rank DP control flow; RCCL needs one GPU per rank)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
W=$(mktemp -d)
python3 - "$R" "$W" <<'PY'
import sys, yaml, os
R, W = sys.argv[1], sys.argv[2]
c = yaml.safe_load(open(os.path.join(R, "configs", "interm_8m.yaml")))
c["trainer"].update(max_epochs=2, batch_size=2)
c["parallelism"].update(simple_ddp=2)
c["model"].update(depth=2, warmup_epochs=1)
c["data"]["synthetic"]["ERA5_1"].update(steps_per_epoch=2)
yaml.safe_dump(c, open(os.path.join(W, "dp.yaml"), "w"))
PY
cd $W
for r in 0 1; do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 WORLD_SIZE=2 RANK=$r LOCAL_RANK=0 ORBIT2_DIST_BACKEND=gloo \
    python3 $R/examples/intermediate_downscaling.py dp.yaml > out$r.log 2>&1 &
  pids[$r]=$!
done
rc=0
for r in 0 1; do wait ${pids[$r]} || rc=1; done
tail -n 6 out0.log; echo ---; tail -n 4 out1.log; exit $rc
