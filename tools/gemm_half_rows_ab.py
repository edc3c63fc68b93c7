"""128-wide GEMM kernel on 64-row tiles (tile hint 64: twice the workgroups, two per CU on single-round grids) against its 128 x 128
tiles (hint 128) at the interm_117m shapes whose 128 x 128 grids are one round of one workgroup per CU: bit-equality (same k order
per accumulator; bias + GELU + dropout + residual epilogue and fp32 output), then interleaved timing rounds (one process, one box).
argv: tokens (default 4096 = interm_117m at batch 8 on the 32 x 64 grid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ok = True
for (M, N, K) in ((192, 256, 128), (4096, 1024, 1024), (1000, 384, 200), (4096, 1024, 4096)):
    for fname, b_kc in (("NT", True), ("NN", False)):
        A, W = r(M, K), (r(N, K) if b_kc else r(K, N))
        ldb = K if b_kc else N
        b, res = r(N), r(M, N)
        outs, o32 = [], []
        for tile in (128, 64):
            o = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
            pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            _hip.gemm(A, W, o, M, N, K, K, ldb, N, a_kc=True, b_kc=b_kc, bias=b, act=1, save_pre=pre, drop_p=0.1, seed=77, residual=res, ldr=N, tile=tile)
            o2 = torch.empty(M, N, dtype=torch.float32, device="cuda")
            _hip.gemm(A, W, o2, M, N, K, K, ldb, N, a_kc=True, b_kc=b_kc, tile=tile)
            outs.append((o, pre)); o32.append(o2)
        torch.cuda.synchronize()
        ref = A.float() @ (W.float().t() if b_kc else W.float())
        same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(o32[0], o32[1])
        e = float((o32[1] - ref).abs().max())
        print("check %s M=%d N=%d K=%d : 64-row == 128-row bitwise %s, fp32 max err %.2e" % (fname, M, N, K, same, e), flush=True)
        ok = ok and same and e < 0.5
print("ALL OK" if ok else "MISMATCH", flush=True)
if not ok:
    sys.exit(1)

def t(f, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def bench(name, M, N, K, b_kc):
    A, W = r(M, K), (r(N, K) if b_kc else r(K, N))
    ldb = K if b_kc else N
    b = r(N)
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    best = {128: [], 64: []}
    for rnd in range(5):
        for tile in (128, 64):
            f = lambda: _hip.gemm(A, W, o, M, N, K, K, ldb, N, a_kc=True, b_kc=b_kc, bias=b, tile=tile)
            if rnd == 0: f()
            best[tile].append(t(f))
    fl = 2.0 * M * N * K / 1e6
    a, c = sorted(best[128])[2], sorted(best[64])[2]
    print("%-12s M=%5d N=%5d K=%5d | 128-row %7.1f us %5.0f TF | 64-row %7.1f us %5.0f TF | %+.1f %%" % (name, M, N, K, a, fl / a, c, fl / c, 100.0 * (a / c - 1)), flush=True)
D = 1024
for name, N, K in (("NT proj", D, D), ("NT fc2", D, 4 * D), ("NT head", D, D)):
    bench(name, T, N, K, True)
for name, N, K in (("NN dX proj", D, D), ("NN dX qkv", D, 3 * D), ("NN dX fc1", D, 4 * D)):
    bench(name, T, N, K, False)
for name, N, K in (("NT 2048 tok", D, 4 * D),):
    bench(name, T // 2, N, K, True)
