"""NT vs TN on the ring (256) and 128 kernels, one quantisation-free shape, for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
from tools.gemm_bench import bench
for form in ("nt", "tn"):
    for tile in (256, 128):
        ms, tf = bench(12288, 12288, 16384, form, tile, iters=3)
        print(form, tile, round(ms, 3), round(tf), flush=True)
