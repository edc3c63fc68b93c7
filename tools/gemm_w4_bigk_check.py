import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
for (M, N, K, lda, ldb) in ((256, 256, 131072, 12352, 3072), (256, 512, 131072, 9216, 3072), (256, 256, 65536, 12352, 3072), (256, 256, 262144, 12352, 3072)):
    A, W = r(K, lda), r(K, ldb)
    o = [torch.empty(M, N, dtype=torch.float32, device="cuda") for _ in range(2)]
    for i, tile in enumerate((256, 260)):
        _hip.gemm(A, W, o[i], M, N, K, lda, ldb, N, a_kc=False, b_kc=False, tile=tile)
    torch.cuda.synchronize()
    print("TN M=%d N=%d K=%d lda=%d: equal %s, max |diff| %.3e, max |ref| %.3e" % (M, N, K, lda, torch.equal(o[0], o[1]), float((o[0] - o[1]).abs().max()), float(o[0].abs().max())), flush=True)
