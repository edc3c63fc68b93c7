"""Does the 4-wave kernel's LDS-DMA issue cost depend on the operand's row stride (pages touched per piece)?  Same M, N, K,
A taken as a column slice of wider matrices (lda = K, 2K, 4K, 4K + 64, 8K).  Hints from argv (default 256,260)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
def t(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
hints = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256,260").split(",")]
M, N, K = 65536, 3072, 3072
W, b = r(N, K), r(N)
o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for lda in (K, 2 * K, 4 * K, 4 * K + 64, 8 * K):
    Abig = r(M, lda)
    res = {v: [] for v in hints}
    for rnd in range(3):
        for v in hints:
            f = lambda: _hip.gemm(Abig, W, o, M, N, K, lda, K, N, bias=b, tile=v)
            if rnd == 0: f()
            res[v].append(t(f))
    fl = 2.0 * M * N * K / 1e9
    print("lda=%6d | " % lda + " | ".join("%d: %6.3f ms %5.0f TF" % (v, sorted(x)[1], fl / sorted(x)[1]) for v, x in res.items()), flush=True)
