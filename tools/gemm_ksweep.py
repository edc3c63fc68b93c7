"""Per-tile fixed cost (prologue + epilogue) of the 256-tile kernels: time of [M, N] x K for a sweep of K, linear fit
t = rounds * (t_fixed + nk * t_ktile).  argv: comma list of tile hints (default 256,260)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
def t(f, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
hints = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256,260").split(",")]
M, N = 65536, 9216
rounds = (M // 256) * (N // 256) / 256.0
Ks = (128, 256, 512, 1024, 2048, 3072)
for bias in (True,):
    for v in hints:
        ts = []
        for K in Ks:
            A, W, b = r(M, K), r(N, K), r(N)
            o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            f = lambda: _hip.gemm(A, W, o, M, N, K, K, K, N, bias=b if bias else None, tile=v)
            f(); f()
            ts.append(sorted(t(f) for _ in range(3))[1])
        # least squares on (nk, t)
        nk = [K // 64 for K in Ks]
        n = len(nk); sx = sum(nk); sy = sum(ts); sxx = sum(x * x for x in nk); sxy = sum(x * y for x, y in zip(nk, ts))
        slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
        print("hint %d: " % v + " ".join("K=%d %.3f ms" % (K, x) for K, x in zip(Ks, ts)), flush=True)
        print("   per tile: fixed %.2f us, per K-tile %.3f us (= %.0f TF in the loop); output %.2f GB -> %.2f TB/s if the fixed part were only the store"
              % (icpt / rounds * 1e3, slope / rounds * 1e3, 2.0 * 256 * 256 * 64 * 256 / (slope / rounds * 1e-3) / 1e12, M * N * 2 / 1e9, M * N * 2 / 1e9 / icpt), flush=True)
