"""Times the eight K-contiguous GEMMs of one interm_1b Block (forward + input-gradient) with their real epilogues."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

BF = torch.bfloat16
def t(f, n=6):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L, D, hid = 8192, 3072, 12288
M = B * L
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(BF)
x, xh, x3 = r(M, D), r(M, hid), r(M, 3 * D)
wqkv, wp, w1, w2 = r(3 * D, D), r(D, D), r(hid, D), r(D, hid)
bq, bp, b1, b2 = r(3 * D), r(D), r(hid), r(D)
res = r(M, D)
dp = torch.full((B,), 1.0 / 0.9, device="cuda")
pre = torch.empty(M, hid, dtype=BF, device="cuda")
oD, oH, o3 = torch.empty(M, D, dtype=BF, device="cuda"), torch.empty(M, hid, dtype=BF, device="cuda"), torch.empty(M, 3 * D, dtype=BF, device="cuda")
cases = [
    ("qkv fwd  +bias", lambda kw: _hip.gemm(x, wqkv, o3, M, 3 * D, D, D, D, 3 * D, bias=bq, **kw), 3 * D, D),
    ("proj fwd +bias drop rowscale res", lambda kw: _hip.gemm(x, wp, oD, M, D, D, D, D, D, bias=bp, drop_p=0.1, seed=5, rowscale=dp, rows_per_scale=L, residual=res, ldr=D, **kw), D, D),
    ("fc1 fwd  +bias gelu save_pre drop", lambda kw: _hip.gemm(x, w1, oH, M, hid, D, D, D, hid, bias=b1, act=1, save_pre=pre, drop_p=0.1, seed=6, **kw), hid, D),
    ("fc2 fwd  +bias drop rowscale res", lambda kw: _hip.gemm(xh, w2, oD, M, D, hid, hid, hid, D, bias=b2, drop_p=0.1, seed=7, rowscale=dp, rows_per_scale=L, residual=res, ldr=D, **kw), D, hid),
    ("fc2 dX   drop dgelu", lambda kw: _hip.gemm(x, w1, oH, M, hid, D, D, D, hid, drop_p=0.1, seed=6, dgelu_pre=pre, **kw), hid, D),
    ("fc1 dX   plain", lambda kw: _hip.gemm(xh, w2, oD, M, D, hid, hid, hid, D, **kw), D, hid),
    ("proj dX  plain", lambda kw: _hip.gemm(x, wp, oD, M, D, D, D, D, D, **kw), D, D),
    ("qkv dX   plain (K=3D)", lambda kw: _hip.gemm(x3, w2[:, :3 * D].contiguous(), oD, M, D, 3 * D, 3 * D, 3 * D, D, **kw), D, 3 * D),
    ("fc1 shape plain", lambda kw: _hip.gemm(x, w1, oH, M, hid, D, D, D, hid, **kw), hid, D),
    ("qkv shape plain", lambda kw: _hip.gemm(x, wqkv, o3, M, 3 * D, D, D, D, 3 * D, **kw), 3 * D, D),
]
hints = [int(h) for h in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]     # e.g. 256,260: 8-phase against 4-wave kernel
tot = {h: 0.0 for h in hints}
for name, f, N, K in cases:
    fl = 2.0 * M * N * K
    ms = {h: [] for h in hints}
    for rnd in range(3):
        for h in hints:
            ms[h].append(t(lambda: f({"tile": h} if h else {})))
    line = "%-36s N=%6d K=%6d" % (name, N, K)
    for h in hints:
        m = sorted(ms[h])[1]
        if "shape" not in name: tot[h] += m
        line += " | [%d] %7.3f ms %6.0f TF" % (h, m, fl / m / 1e9)
    print(line, flush=True)
for h in hints:
    print("[%d] block NT total %.3f ms (x8 blocks = %.1f ms/step)" % (h, tot[h], 8 * tot[h]))
