"""fc2-like operands (row stride 12288 for both A and B) at contraction lengths 3072 / 6144 / 12288: does the 4-wave kernel's
loop slow down with the LENGTH of the K loop (CUs of an XCD drifting apart -> panel lines no longer shared through L2)?"""
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
M, N = 65536, 3072
LD = int(sys.argv[2]) if len(sys.argv) > 2 else 12288
A, W, b = r(M, LD), r(N, LD), r(N)
o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for K in (3072, 6144, 12288):
    res = {v: [] for v in hints}
    for rnd in range(3):
        for v in hints:
            f = lambda: _hip.gemm(A, W, o, M, N, K, LD, LD, N, bias=b, tile=v)
            if rnd == 0: f()
            res[v].append(t(f))
    fl = 2.0 * M * N * K / 1e9
    print("K=%6d (ld %d) | " % (K, LD) + " | ".join("%d: %6.3f ms %5.0f TF" % (v, sorted(x)[1], fl / sorted(x)[1]) for v, x in res.items()), flush=True)
