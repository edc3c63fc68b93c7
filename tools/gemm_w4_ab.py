"""4-wave 256x256x64 kernel (tile hint 260: one wave per SIMD, 128x128 per wave, hand-placed main loop) against the 8-phase
8-wave kernel (256) in every operand form: bit-equality (same k order per accumulator; generic epilogue with GELU + dropout +
residual, fp32 output), then interleaved timing rounds on the Block's shapes (one process, one box).
argv: tokens (default 65536 = batch 8), 'fast' skips the checks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
FORMS = (("NT", True, True), ("NN", True, False), ("TN", False, False), ("TT", False, True))
ok = True
if "fast" not in sys.argv:
    for (M, N, K) in ((256, 256, 64), (256, 512, 128), (512, 768, 192), (1024, 512, 1024), (2048, 3072, 3072)):
        for fname, a_kc, b_kc in FORMS:
            A = r(M, K) if a_kc else r(K, M)
            W = r(N, K) if b_kc else r(K, N)
            lda, ldb = (K if a_kc else M), (K if b_kc else N)
            b, res = r(N), r(M, N)
            pre = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
            outs, o32, ob = [], [], []
            for i, tile in enumerate((256, 260)):
                o = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
                _hip.gemm(A, W, o, M, N, K, lda, ldb, N, a_kc=a_kc, b_kc=b_kc, bias=b, act=1, save_pre=pre[i], drop_p=0.1, seed=1234, residual=res, ldr=N, tile=tile)
                outs.append(o)
                o2 = torch.empty(M, N, dtype=torch.float32, device="cuda")
                _hip.gemm(A, W, o2, M, N, K, lda, ldb, N, a_kc=a_kc, b_kc=b_kc, tile=tile)
                o32.append(o2)
                o3 = (torch.ones(M, N, device="cuda") * 0.25).to(torch.bfloat16)
                _hip.gemm(A, W, o3, M, N, K, lda, ldb, N, a_kc=a_kc, b_kc=b_kc, bias=b, beta=1.0, tile=tile)   # lean path + beta -> generic
                ob.append(o3)
            torch.cuda.synchronize()
            ref = (A.float() if a_kc else A.float().t()) @ (W.float().t() if b_kc else W.float())
            e = float((o32[1] - ref).abs().max())
            same = torch.equal(outs[0], outs[1]) and torch.equal(pre[0], pre[1]) and torch.equal(o32[0], o32[1]) and torch.equal(ob[0], ob[1])
            print("check %s M=%d N=%d K=%d : 260 == 256 bitwise %s, fp32 max err %.2e" % (fname, M, N, K, same, e), flush=True)
            if not torch.equal(o32[0], o32[1]):
                bad = (o32[0] != o32[1]).nonzero()
                print("   %d differ; first (m, n): %s; row blocks %s; col blocks %s" % (len(bad), bad[:6].tolist(), sorted(set((bad[:, 0] // 16).tolist()))[:20], sorted(set((bad[:, 1] // 16).tolist()))[:20]), flush=True)
            ok = ok and same and e < 0.5
    print("ALL OK" if ok else "MISMATCH", flush=True)
    if not ok:
        sys.exit(1)

def t(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
args = [a for a in sys.argv[1:] if a != "fast"]
T, D = (int(args[0]) if args else 65536), 3072
def bench(name, M, N, K, a_kc, b_kc, bias=True):
    A = r(M, K) if a_kc else r(K, M)
    W = r(N, K) if b_kc else r(K, N)
    lda, ldb = (K if a_kc else M), (K if b_kc else N)
    b = r(N) if bias else None
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    best = {256: [], 260: []}
    for rnd in range(4):
        for tile in (256, 260):
            f = lambda: _hip.gemm(A, W, o, M, N, K, lda, ldb, N, a_kc=a_kc, b_kc=b_kc, bias=b, tile=tile)
            if rnd == 0: f()
            best[tile].append(t(f))
    fl = 2.0 * M * N * K / 1e9
    m6, m7 = sorted(best[256])[len(best[256]) // 2], sorted(best[260])[len(best[260]) // 2]
    print("%-9s M=%6d N=%6d K=%6d | phase8 %7.3f ms %5.0f TF | w4 %7.3f ms %5.0f TF | %+.1f %%" % (name, M, N, K, m6, fl / m6, m7, fl / m7, 100.0 * (m6 / m7 - 1)), flush=True)
for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
    bench("NT " + name, T, N, K, True, True)
for name, No, Ki in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
    bench("NN dX" + name, T, Ki, No, True, False, bias=False)       # dX[T, K_in] = dY[T, N_out] . W[N_out, K_in]
for name, No, Ki in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
    bench("TN dW" + name, No, Ki, T, False, False, bias=False)      # dW[N_out, K_in] = dY^T . X
