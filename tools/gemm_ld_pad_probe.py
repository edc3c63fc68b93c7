"""Row-stride (leading dimension) sensitivity of the 256-tile kernels on the MLP shapes: operands taken as column slices of
wider allocations so that lda / ldb are K or K + 64 (+128 bytes: breaks the power-of-two row stride that puts the same k-offset
of every row on the same memory channel)."""
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
M = 65536
for name, N, K in (("fc2", 3072, 12288), ("fc1", 12288, 3072), ("qkv", 9216, 3072)):
    b = r(N)
    for pa, pb, pc in ((0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (0, 0, 64), (64, 64, 64)):
        A, W = r(M, K + pa), r(N, K + pb)
        o = torch.empty(M, N + pc, dtype=torch.bfloat16, device="cuda")
        res = {v: [] for v in hints}
        for rnd in range(3):
            for v in hints:
                f = lambda: _hip.gemm(A, W, o, M, N, K, K + pa, K + pb, N + pc, bias=b, tile=v)
                if rnd == 0: f()
                res[v].append(t(f))
        fl = 2.0 * M * N * K / 1e9
        print("%-4s lda=K+%-2d ldb=K+%-2d ldc=N+%-2d | " % (name, pa, pb, pc) + " | ".join("%d: %6.3f ms %5.0f TF" % (v, sorted(x)[1], fl / sorted(x)[1]) for v, x in res.items()), flush=True)
