import os, sys, json, subprocess
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
sys.argv = ["bench.py", "--model", "interm_117m", "--grid", "32x64", "--batch", "8", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--graph", "off"]
import runpy
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    try:
        runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
    except SystemExit:
        pass
ev = [e for e in prof.events() if e.name.startswith("aten::") and e.name in ("aten::fill_", "aten::copy_", "aten::zero_", "aten::_to_copy", "aten::clone", "aten::cat", "aten::mul", "aten::add", "aten::div", "aten::sub", "aten::select", "aten::index", "aten::zeros", "aten::full")]
import collections
c = collections.Counter()
for e in ev:
    st = [s for s in (e.stack or []) if "orbit-2_amd" in s or "bench.py" in s or "oracle" in s]
    key = (e.name, str(e.input_shapes)[:60], st[0][-70:] if st else "?")
    c[key] += 1
for k, v in c.most_common(60):
    print(v, k)
