#!/bin/bash
# same-box A/B of the attention kernels: tools/ab_attn.sh <alt-lib-name>   (rocprofv3 per-kernel averages, both builds)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in base alt base2; do
  lib=""; [ "$tag" = alt ] && lib="$R/orbit-2_amd/lib/alt/$1.so"
  ORBIT2_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$tag -o a -- python3 $R/tools/attn_prof.py > $R/gpurun_out/ab_$tag.log 2>&1
  (cd $R && python tools/summarize_prof.py stats gpurun_out/ab_$tag/a_kernel_stats.csv gpurun_out/ab_$tag.txt)
done
cd $R
python - <<'PY'
import re
def load(p):
    d={}
    for l in open(p):
        m=re.match(r"(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+%",l)
        if m and "attn" in m.group(1): d[m.group(1).strip()]=float(m.group(4))
    return d
a,b,c=load("gpurun_out/ab_base.txt"),load("gpurun_out/ab_alt.txt"),load("gpurun_out/ab_base2.txt")
def norm(k): return re.sub(r", (true|false)>$", ">", k) if k.count(",")>=2 and "dkv" not in k else k
bb={norm(k):v for k,v in b.items()}
bb.update({re.sub(r", false>$", ">", k):v for k,v in b.items()})
for k in sorted(a):
    alt=b.get(k, bb.get(norm(k), bb.get(re.sub(r", false>$", ">", k), float("nan"))))
    print("%-48s base %8.1f  alt %8.1f  base-again %8.1f  alt/base %.3f" % (k, a[k], alt, c.get(k,float('nan')), alt/((a[k]+c.get(k,a[k]))/2)))
PY
