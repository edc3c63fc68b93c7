"""Kernel-rate comparison of the four operand forms on a quantisation-free shape (48x48 tiles of 256, 96x96 of 128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
from tools.gemm_bench import bench
for K in (4096, 16384):
    for form in ("nt", "nn", "tn", "tt"):
        r = {t: bench(12288, 12288, K, form, t, iters=5) for t in (128, 256)}
        print("M=N=12288 K=%5d %s | 128: %7.3f ms %6.0f TF | 256: %7.3f ms %6.0f TF" % (K, form, r[128][0], r[128][1], r[256][0], r[256][1]), flush=True)
