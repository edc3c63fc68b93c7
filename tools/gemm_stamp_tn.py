"""cycle stamps of the 8-phase kernel with contiguous units on a weight-gradient shape (diagnostic library)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import numpy as np
import torch
from climate_learn import _hip
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
T = 65536
for name, Mo, No, akc, bkc in (("dW fc1 (TN)", 12288, 3072, False, False), ("dX fc1 (NN)", T, 3072, True, False)):
    if akc:
        K = 12288
        A, B = r(Mo, K), r(K, No)
        lda, ldb = K, No
    else:
        K = T
        A, B = r(K, Mo), r(K, No)
        lda, ldb = Mo, No
    o = torch.empty(Mo, No, dtype=torch.bfloat16, device="cuda")
    for _ in range(2):
        _hip.gemm(A, B, o, Mo, No, K, lda, ldb, No, a_kc=akc, b_kc=bkc, tile=258)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _hip.gemm(A, B, o, Mo, No, K, lda, ldb, No, a_kc=akc, b_kc=bkc, tile=258); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = (C.c_uint * (64 * 2 * 8))()
    _hip.lib().orbit2_debug_read(buf, 64 * 2 * 8)
    d = np.frombuffer(buf, dtype=np.uint32).reshape(64, 2, 8).astype(np.float64)
    nk = d[0, 0, 6]
    print("%s M=%d N=%d K=%d  %.3f ms  %.0f TF (stamped build) nk=%d kernel id %d" % (name, Mo, No, K, ms, 2.0 * Mo * No * K / ms / 1e9, nk, d[0, 0, 7]))
    for g, nm in ((0, "waves0-3"), (1, "waves4-7")):
        v = d[:, g, :6].mean(0)
        print("   %s per 64-deep K-tile (4 phases): L %.0f | barrier_a %.0f | M %.0f | vmcnt %.0f | barrier_b %.0f | sum %.0f"
              % (nm, v[0] / nk, v[1] / nk, v[2] / nk, v[3] / nk, v[4] / nk, v[:5].sum() / nk))
