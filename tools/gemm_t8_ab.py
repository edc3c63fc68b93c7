"""8-phase kernel with contiguous units (tile hint 258; all operand forms) against the 128-tile kernel: bit-equality in every
form (same k order per accumulator), grouped launch, then interleaved timing on the Block's weight-gradient (TN) shapes
and input-gradient (NN on the stored weight vs NT on a transposed copy) shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
ok = True
for (M, N, K) in ((256, 256, 128), (512, 768, 128), (1000, 520, 192), (3072, 3072, 4096), (776, 1032, 1024)):
    for a_kc, b_kc in ((True, True), (True, False), (False, True), (False, False)):
        A = r(M, K) if a_kc else r(K, M)
        B = r(N, K) if b_kc else r(K, N)
        lda, ldb = (K if a_kc else M), (K if b_kc else N)
        outs = []
        for tile in (128, 258):
            o = (torch.ones(M, N, device="cuda") * 0.25).to(torch.bfloat16)
            _hip.gemm(A, B, o, M, N, K, lda, ldb, N, a_kc=a_kc, b_kc=b_kc, beta=1.0, tile=tile)
            outs.append(o)
        o32 = [torch.empty(M, N, dtype=torch.float32, device="cuda") for _ in range(2)]
        for i, tile in enumerate((128, 258)):
            _hip.gemm(A, B, o32[i], M, N, K, lda, ldb, N, a_kc=a_kc, b_kc=b_kc, tile=tile)
        torch.cuda.synchronize()
        ref = (A.float() if a_kc else A.float().t()) @ (B.float().t() if b_kc else B.float())
        e = float((o32[1] - ref).abs().max())
        same = torch.equal(outs[0], outs[1]) and torch.equal(o32[0], o32[1])
        print("check M=%d N=%d K=%d a_kc=%d b_kc=%d : 258 == 128 bitwise %s, fp32 max err %.2e" % (M, N, K, a_kc, b_kc, same, e), flush=True)
        ok = ok and same and e < 0.5
# grouped launch, TN with accumulation: 256-tile group == per-problem results
K = 2048
probs, refs = [], []
for (M, N) in ((512, 768), (1024, 512), (768, 768), (1280, 256)):
    A, B = r(K, M), r(K, N)
    o1 = (torch.ones(M, N, device="cuda") * 0.5).to(torch.bfloat16)
    o2 = o1.clone()
    _hip.gemm(A, B, o1, M, N, K, M, N, N, a_kc=False, b_kc=False, beta=1.0, tile=128)
    probs.append((A, B, o2, M, N, K, M, N, N, dict(a_kc=False, b_kc=False, beta=1.0, tile=258)))
    refs.append(o1)
_hip.gemm_grouped(probs)
torch.cuda.synchronize()
g_ok = all(torch.equal(p[2], q) for p, q in zip(probs, refs))
print("grouped 256-tile launch == per-problem 128-tile results: %s" % g_ok, flush=True)
ok = ok and g_ok
print("ALL OK" if ok else "MISMATCH", flush=True)
if not ok:
    sys.exit(1)

def t(f, n=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
med = lambda v: sorted(v)[len(v) // 2]
T, D = (int(sys.argv[1]) if len(sys.argv) > 1 else 65536), 3072      # tokens (batch 8)
# ---- weight gradients: dW[N_out, K_in] = dY^T . X, one by one and as the Block's grouped launch
shapes = (("dW qkv", 3 * D, D), ("dW proj", D, D), ("dW fc1", 4 * D, D), ("dW fc2", D, 4 * D))
ops = {}
for name, Mo, No in shapes:
    ops[name] = (r(T, Mo), r(T, No), torch.empty(Mo, No, dtype=torch.bfloat16, device="cuda"))
for name, Mo, No in shapes:
    dy, x, o = ops[name]
    res = {128: [], 258: []}
    for rnd in range(3):
        for tile in (128, 258):
            res[tile].append(t(lambda: _hip.gemm(dy, x, o, Mo, No, T, Mo, No, No, a_kc=False, b_kc=False, tile=tile)))
    f = 2.0 * Mo * No * T / 1e9
    a, b = med(res[128]), med(res[258])
    print("%-8s M=%6d N=%6d K=%6d | tile128 %7.3f ms %5.0f TF | phase8 %7.3f ms %5.0f TF | %+.1f %%" % (name, Mo, No, T, a, f / a, b, f / b, 100 * (a / b - 1)), flush=True)
res = {128: [], 258: []}
ftot = sum(2.0 * Mo * No * T for _, Mo, No in shapes) / 1e9
for rnd in range(3):
    for tile in (128, 258):
        pr = [(ops[n][0], ops[n][1], ops[n][2], Mo, No, T, Mo, No, No, dict(a_kc=False, b_kc=False, tile=tile)) for n, Mo, No in shapes]
        res[tile].append(t(lambda: _hip.gemm_grouped(pr)))
a, b = med(res[128]), med(res[258])
print("grouped 4 dW               | tile128 %7.3f ms %5.0f TF | phase8 %7.3f ms %5.0f TF | %+.1f %%" % (a, ftot / a, b, ftot / b, 100 * (a / b - 1)), flush=True)
# ---- input gradients: dX[T, K_in] = dY[T, N_out] . W[N_out, K_in]: NN on the stored weight (258) vs NT on a transposed copy (257)
for name, No, Ki in (("dX qkv", 3 * D, D), ("dX proj", D, D), ("dX fc1", 4 * D, D), ("dX fc2", D, 4 * D)):
    dy, W = r(T, No), r(No, Ki)
    Wt = W.t().contiguous()
    o = torch.empty(T, Ki, dtype=torch.bfloat16, device="cuda")
    res = {"nt257": [], "nn258": [], "nn128": []}
    for rnd in range(3):
        res["nt257"].append(t(lambda: _hip.gemm(dy, Wt, o, T, Ki, No, No, No, Ki, tile=257)))
        res["nn258"].append(t(lambda: _hip.gemm(dy, W, o, T, Ki, No, No, Ki, Ki, a_kc=True, b_kc=False, tile=258)))
        res["nn128"].append(t(lambda: _hip.gemm(dy, W, o, T, Ki, No, No, Ki, Ki, a_kc=True, b_kc=False, tile=128)))
    f = 2.0 * T * Ki * No / 1e9
    print("%-8s M=%6d N=%6d K=%6d | NT(257, W^T copy) %7.3f ms %5.0f TF | NN(258) %7.3f ms %5.0f TF | NN(128) %7.3f ms %5.0f TF" %
          (name, T, Ki, No, med(res["nt257"]), f / med(res["nt257"]), med(res["nn258"]), f / med(res["nn258"]), med(res["nn128"]), f / med(res["nn128"])), flush=True)
# ---- forward (NT) shapes: interleaved-unit kernel (257: reads retired before the barrier) vs contiguous-unit kernel (258)
for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
    A, W, b = r(T, K), r(N, K), r(N)
    o = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
    res = {257: [], 258: []}
    for rnd in range(4):
        for tile in (257, 258):
            res[tile].append(t(lambda: _hip.gemm(A, W, o, T, N, K, K, K, N, bias=b, tile=tile)))
    f = 2.0 * T * N * K / 1e9
    a, b_ = med(res[257]), med(res[258])
    print("%-8s M=%6d N=%6d K=%6d | NT 257 %7.3f ms %5.0f TF | NT 258 %7.3f ms %5.0f TF | %+.1f %%" % (name, T, N, K, a, f / a, b_, f / b_, 100 * (a / b_ - 1)), flush=True)
