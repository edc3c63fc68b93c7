"""Row-pitch sensitivity of the 4-wave GEMM at the interm_10b widths (D = 8192: rows 16 KiB apart = one memory-channel period):
the A operand already padded (what round 4 built), then the weight pitch (ldb) and the output pitch (ldc) padded in turn --
what the un-padded weights and Block-boundary tensors still cost.  NT (forward) and NN (input gradient) forms, batch 2 (16384 tokens)."""
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
M = 16384
for name, N, K, nt in (("proj fwd NT", 8192, 8192, True), ("fc1 fwd NT ", 32768, 8192, True), ("proj dX NN ", 8192, 8192, False), ("fc1 dX NN  ", 8192, 32768, False)):
    rows = []
    for pa, pb, pc in ((0, 0, 0), (64, 0, 0), (64, 64, 0), (64, 0, 64), (64, 64, 64)):
        A = r(M, K + pa)
        W = r(N, K + pb) if nt else r(K, N + pb)
        o = torch.empty(M, N + pc, dtype=torch.bfloat16, device="cuda")
        f = lambda: _hip.gemm(A, W, o, M, N, K, K + pa, (K if nt else N) + pb, N + pc, a_kc=True, b_kc=nt, tile=260)
        f()
        ts = sorted(t(f) for _ in range(3))
        rows.append("lda+%-2d ldb+%-2d ldc+%-2d %6.3f ms %5.0f TF" % (pa, pb, pc, ts[1], 2.0 * M * N * K / ts[1] / 1e9))
    print(name + " | " + " | ".join(rows), flush=True)
