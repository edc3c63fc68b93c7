#!/bin/bash
# per-kernel average durations of the attention kernels for several builds of the library (rocprofv3 --kernel-trace --stats):
#   bash tools/attn_kstats.sh "B H L d" lib1.so lib2.so ...     (libs relative to the repo root) -> gpurun_out/kstats_<name>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
shape=$1; shift
for lib in "$@"; do
  name=$(basename $lib .so)
  ORBIT2_HIP_LIB=$R/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$name -o a -- python3 $R/tools/attn_prof.py $shape > $R/gpurun_out/ks_$name.log 2>&1
  (cd $R && python tools/summarize_prof.py stats gpurun_out/ks_$name/a_kernel_stats.csv gpurun_out/kstats_$name.txt && echo "== $name ($shape)" && grep "attn" gpurun_out/kstats_$name.txt)
done
