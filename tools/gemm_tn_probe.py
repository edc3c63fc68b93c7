import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
from tools.gemm_bench import bench
for (M, N, K) in [(4096, 4096, 4096), (3072, 3072, 4096), (12288, 3072, 2048), (12288, 3072, 8192), (12288, 3072, 32768)]:
    for form in ("nt", "nn", "tn"):
        r = {t: bench(M, N, K, form, t) for t in (128, 256)}
        print("%s M=%6d N=%6d K=%6d | 128: %6.0f TF | 256: %6.0f TF" % (form, M, N, K, r[128][1], r[256][1]), flush=True)
