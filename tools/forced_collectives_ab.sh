#!/bin/bash
# same-box A/B of the single-rank step: plain / collectives forced on (RCCL, 1 rank) without CommStats / with CommStats,
# two rounds interleaved:   bash tools/forced_collectives_ab.sh  -> gpurun_out/r04_forced_ab.txt
R=${GRAFT_REPO_ROOT:-.}
OUT=$R/gpurun_out/r04_forced_ab.txt
: > $OUT
for round in 1 2; do
  for mode in plain forced_nostats forced_stats; do
    case $mode in
      plain) env="ORBIT2_FORCE_COLLECTIVES=0"; extra="";;
      forced_nostats) env="ORBIT2_FORCE_COLLECTIVES=1"; extra="--no-comm-stats";;
      forced_stats) env="ORBIT2_FORCE_COLLECTIVES=1"; extra="";;
    esac
    env $env MASTER_ADDR=127.0.0.1 MASTER_PORT=29688 python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline $extra 2>/dev/null | tail -n 1 > /tmp/fab.json
    python - >> $OUT <<PY
import json
d = json.loads(open("/tmp/fab.json").read())
c = d.get("comm") or {}
print("round $round  %-16s %8.3f samples/s  %9.2f ms/step   comm_ms %s exposed %s rccl_ranks %s" % ("$mode", d["value"], d["ms_per_step"], c.get("comm_ms_per_step"), c.get("exposed_comm_ms_per_step"), d.get("rccl_ranks")))
PY
  done
done
cat $OUT
