"""Weight-gradient (K-strided) GEMM rate vs contraction length and leading dimension (fc1 dW: 12288 x 3072)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
def t(f, n=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M, N = 12288, 3072
for K in (8192, 16384, 32768, 65536, 131072):
    A = torch.randn(K, M, device="cuda").to(torch.bfloat16)
    B = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ms = t(lambda: _hip.gemm(A, B, out, M, N, K, M, N, N, a_kc=False, b_kc=False, tile=128))
    print("fc1 dW K=%6d | %8.3f ms %6.0f TF" % (K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    del A, B
