"""Reference points only (never a product path): hipBLASLt through torch on the shapes of tools/gemm_w4_probe.py, same random
data, same timing loop.  Run under rocprofv3 --kernel-trace --stats to see which library kernel serves each shape."""
import sys, torch
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
def t(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
D = 3072
for name, Mm, N, K in (("qkv", 65536, 3 * D, D), ("fc2", 65536, D, 4 * D), ("8192^3", 8192, 8192, 8192), ("longK", 4096, 4096, 65536)):
    A, W, b = r(Mm, K), r(N, K), r(N)
    o = torch.empty(Mm, N, dtype=torch.bfloat16, device="cuda")
    for f, tag in ((lambda: torch.mm(A, W.t(), out=o), "mm(A, W^T)"), (lambda: torch.addmm(b, A, W.t(), out=o), "addmm bias")):
        f(); f()
        ts = sorted(t(f) for _ in range(4))
        fl = 2.0 * Mm * N * K / 1e9
        print("%-7s M=%6d N=%6d K=%6d | %-11s %7.3f ms %5.0f TF" % (name, Mm, N, K, tag, ts[1], fl / ts[1]), flush=True)
