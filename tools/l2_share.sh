#!/bin/bash
# on the GPU box: bash tools/l2_share.sh -> gpurun_out/r06_l2_share_probe.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/l2share
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/l2share -o probe -- python3 $R/tools/l2_share_probe.py > $R/gpurun_out/l2share/probe.log 2>&1
cd $R
python3 - <<'PY' > gpurun_out/r06_l2_share_probe.txt
import csv, glob, collections, sys
sys.path.insert(0, "tools")
from l2_share_probe import CASES, KS
f = glob.glob("gpurun_out/l2share/**/probe_counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm256w_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
print("# TN 4-wave GEMM, FETCH_SIZE x 2 x 1024 per launch (last 2 of 4 launches), against what the cohorts would fetch beyond L2")
print("# ideal: every strip K-tile once per XCD cohort; none: every tile fetches its own two strips")
i = 0
for tm, tn in CASES:
    for K in KS:
        grp = rows[i:i + 4]; i += 4
        fb = sum(float(r["Counter_Value"]) for r in grp[2:]) / 2 * 2 * 1024
        ns = sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in grp[2:]) / 2
        tiles = tm * tn
        none = tiles * 2 * 256 * K * 2
        if tiles == 16:
            ideal = 8 * 3 * 256 * K * 2
        else:
            ideal = (tiles // 32) * 12 * 256 * K * 2
        print("tiles %3d x %d  K %6d  %8.3f ms  FETCH %7.2f GB  ideal %6.2f GB  none %6.2f GB  -> %.2f x ideal, %.0f %% of none"
              % (tm, tn, K, ns / 1e6, fb / 1e9, ideal / 1e9, none / 1e9, fb / ideal, 100 * fb / none))
PY
cat gpurun_out/r06_l2_share_probe.txt
