"""The eight K-contiguous GEMMs of a Block (B=4, plain epilogues) for same-box A/B runs of the ring kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
def t(f, n=6):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M, D = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 3072
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
tot = 0.0
for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D), ("dX qkv", D, 3 * D)):
    A, W = r(M, K), r(N, K)
    b = r(N)
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ms = t(lambda: _hip.gemm(A, W, o, M, N, K, K, K, N, bias=b))
    tot += ms
    print("%-7s N=%6d K=%6d | %7.3f ms %6.0f TF" % (name, N, K, ms, 2.0 * M * N * K / ms / 1e9))
print("total %.3f ms" % tot)
