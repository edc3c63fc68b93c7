import os, sys, collections
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
log = collections.Counter()
orig = _hip.gemm
def gemm(A, B, out, M, N, K, lda, ldb, ldc, **kw):
    key = (M, N, K, lda, ldb, ldc, tuple(sorted((k, (v if isinstance(v, (int, float, bool)) else "T")) for k, v in kw.items() if v is not None and k != "seed")))
    log[key] += 1
    return orig(A, B, out, M, N, K, lda, ldb, ldc, **kw)
_hip.gemm = gemm
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
import runpy
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print(v, k)
