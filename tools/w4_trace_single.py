"""Round 6 diagnostic (-DO2_W4_TRACE build): when does each XCD finish its share of a statically partitioned launch?  Single 4-wave
GEMMs at the bench shapes; per XCD (block b runs on XCD b & 7) the time its last workgroup ends, relative to the kernel's span:
what the chip waits for at every kernel end."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
_hip.LIB_PATH = os.environ.get("ORBIT2_TRACE_LIB", os.path.join(ROOT, "orbit-2_amd", "lib", "alt", "w4trace.so"))
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
D, T = 3072, 131072
for name, M, N, K, a_kc, b_kc in (("NT fc1 fwd", T, 4 * D, D, True, True), ("NN dX fc1", T, D, 4 * D, True, False), ("NT qkv", T, 3 * D, D, True, True)):
    A = r(M, K) if a_kc else r(K, M)
    W = r(N, K) if b_kc else r(K, N)
    lda, ldb = (K if a_kc else M), (K if b_kc else N)
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        _hip.gemm(A, W, o, M, N, K, lda, ldb, N, a_kc=a_kc, b_kc=b_kc, tile=260)
    torch.cuda.synchronize()
    n = min(32768, (M // 256) * (N // 256))
    buf = (C.c_ulonglong * (2 * n))()
    _hip.lib().orbit2_debug_read_w4_trace(buf, 2 * n)
    st = [buf[2 * b] for b in range(n)]; en = [buf[2 * b + 1] for b in range(n)]
    t0, t1 = min(st), max(en)
    ends = [max(en[b] for b in range(x, n, 8)) for x in range(8)]
    busy = [sum(en[b] - st[b] for b in range(x, n, 8)) for x in range(8)]
    print("%-10s %d tiles, span %.1f us (main loops only); per XCD: last end before the kernel's end [us]: %s | sum of sweeps relative to the mean: %s | idle at the end: %.2f %% of the chip"
          % (name, n, (t1 - t0) / 100.0, " ".join("%6.1f" % ((t1 - e) / 100.0) for e in ends),
             " ".join("%.3f" % (b_ * 8 / sum(busy)) for b_ in busy), 100.0 * sum(t1 - e for e in ends) / 8 / (t1 - t0)), flush=True)
