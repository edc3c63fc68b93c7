"""Times the 4-wave GEMM kernel (tile hint 260) of SEVERAL builds of the library in one process, interleaved rounds
(schedule variants built by tools/mkvar_w4.sh into orbit-2_amd/lib/alt/):  python tools/gemm_multi_ab.py lib1.so lib2.so ...
The first library is the reference for the bitwise comparison."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
paths = [a for a in sys.argv[1:] if a.endswith(".so")]
libs = [(os.path.basename(p).replace(".so", ""), C.CDLL(os.path.abspath(p))) for p in paths]
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
def t(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
D, T = 3072, (131072 if "--b16" in sys.argv else 65536)
cases = (("NT qkv", T, 3 * D, D, True, True), ("NT proj", T, D, D, True, True), ("NN dXqkv", T, D, 3 * D, True, False),
         ("NN dXfc1", T, D, 4 * D, True, False), ("NN dXproj", T, D, D, True, False), ("TN dWfc1", 4 * D, D, T, False, False),
         ("TN dWqkv", 3 * D, D, T, False, False))
tot = {n: 0.0 for n, _ in libs}
for name, M, N, K, a_kc, b_kc in cases:
    A = r(M, K) if a_kc else r(K, M)
    W = r(N, K) if b_kc else r(K, N)
    lda, ldb = (K if a_kc else M), (K if b_kc else N)
    outs = {n: torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for n, _ in libs}
    res = {n: [] for n, _ in libs}
    for rnd in range(4):
        for n, lib in libs:
            _hip._lib = lib
            f = lambda: _hip.gemm(A, W, outs[n], M, N, K, lda, ldb, N, a_kc=a_kc, b_kc=b_kc, tile=260)
            if rnd == 0: f()
            res[n].append(t(f))
    torch.cuda.synchronize()
    fl = 2.0 * M * N * K / 1e9
    ref = sorted(res[libs[0][0]])[1]
    line = "%-9s" % name
    for n, _ in libs:
        m = sorted(res[n])[1]
        tot[n] += m
        line += " | %s %6.3f ms %5.0f TF %+5.1f%% %s" % (n, m, fl / m, 100 * (ref / m - 1), "=" if torch.equal(outs[n], outs[libs[0][0]]) else "DIFF")
    print(line, flush=True)
print("sum: " + " | ".join("%s %.3f ms (%+.1f %%)" % (n, tot[n], 100 * (tot[libs[0][0]] / tot[n] - 1)) for n, _ in libs))
