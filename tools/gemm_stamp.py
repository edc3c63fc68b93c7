"""Where does the ring GEMM's main loop spend its cycles?  Needs the diagnostic library (tools/stamp_build.sh):
   ORBIT2_HIP_LIB=orbit-2_amd/lib/alt/stamp.so python tools/gemm_stamp.py
Prints, per wave group (waves 0-3 = A, 4-7 = B), the average shader cycles per K-slab spent in: L (LDS-DMA issue + fragment
reads + lgkmcnt wait), barrier a, M (32 MFMAs + 2 LDS-DMA), vmcnt wait, barrier b."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import numpy as np
import torch
from climate_learn import _hip
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
TILE = int(os.environ.get("TILE", "256"))
for name, N, K in (("qkv", 9216, 3072), ("proj", 3072, 3072), ("fc2", 3072, 12288)):
    A, W = r(M, K), r(N, K)
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        _hip.gemm(A, W, o, M, N, K, K, K, N, tile=TILE)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _hip.gemm(A, W, o, M, N, K, K, K, N, tile=TILE); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = (C.c_uint * (64 * 2 * 8))()
    rc = _hip.lib().orbit2_debug_read(buf, 64 * 2 * 8)
    d = np.frombuffer(buf, dtype=np.uint32).reshape(64, 2, 8).astype(np.float64)
    nk = d[0, 0, 6]
    print("tile %d: %-5s M=%d N=%d K=%d  %.3f ms  %.0f TF (stamped build)  nk=%d" % (TILE, name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, nk))
    for g, nm in ((0, "A waves0-3"), (1, "B waves4-7")):
        v = d[:, g, :6].mean(0)
        print("   %s per K-step (32 deep for tile 256: one L/M pair; 64 deep for 257: four phases): L %.0f | barrier_a %.0f | M %.0f | vmcnt %.0f | barrier_b %.0f | sum %.0f ; whole loop/nk %.0f"
              % (nm, v[0] / nk, v[1] / nk, v[2] / nk, v[3] / nk, v[4] / nk, v[:5].sum() / nk, v[5] / nk))
