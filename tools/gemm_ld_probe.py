"""Does the leading dimension (row stride) of the K-strided operands change TN/NN speed? (channel-camping probe)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

def run(M, N, K, form, pa, pb, tile=128, iters=10):
    a_kc = form[0] == "n"; b_kc = form[1] == "t"
    A = torch.randn((M, K + pa) if a_kc else (K, M + pa), device="cuda").to(torch.bfloat16)
    B = torch.randn((N, K + pb) if b_kc else (K, N + pb), device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    f = lambda: _hip.gemm(A, B, out, M, N, K, A.shape[1], B.shape[1], N, a_kc=a_kc, b_kc=b_kc, tile=tile)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9

for name, M, N, K, form in [("fc1 dW tn", 12288, 3072, 16384, "tn"), ("fc2 dX nn", 16384, 12288, 3072, "nn"),
                            ("fc1 fwd nt", 16384, 12288, 3072, "nt"), ("4096 tn", 4096, 4096, 4096, "tn")]:
    for pa, pb in [(0, 0), (64, 64), (128, 128), (72, 72), (264, 264)]:
        print("%-10s pad(%3d,%3d): %6.0f TF" % (name, pa, pb, run(M, N, K, form, pa, pb)), flush=True)
