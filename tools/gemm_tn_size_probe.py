"""K-strided (tn) vs K-contiguous (nt) ring kernel on one round of 256 tiles: footprint sweep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
from tools.gemm_bench import bench
for (M, N, K) in ((4096, 4096, 4096), (4096, 4096, 32768), (4096, 4096, 131072), (16384, 16384, 2048), (16384, 16384, 16384)):
    r = {(f, t): bench(M, N, K, f, t, iters=4) for f in ("nt", "tn") for t in (256, 128)}
    print("M=%5d N=%5d K=%6d (A+B = %5.0f MB) | nt256 %5.0f  tn256 %5.0f | nt128 %5.0f  tn128 %5.0f TF" %
          (M, N, K, (M + N) * K * 2 / 1e6, r[("nt", 256)][1], r[("tn", 256)][1], r[("nt", 128)][1], r[("tn", 128)][1]), flush=True)
