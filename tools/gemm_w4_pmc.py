"""Workload for a PMC pass (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace): the 8-phase (256) and the
4-wave kernel (tile hint 260) on the qkv and fc2 forward shapes at batch 8, 6 launches each after warm-up."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
D = 3072
for name, Mm, N, K in (("qkv", 65536, 3 * D, D), ("fc2", 65536, D, 4 * D)):
    A, W, b = r(Mm, K), r(N, K), r(N)
    o = torch.empty(Mm, N, dtype=torch.bfloat16, device="cuda")
    for v in (256, 260):
        for _ in range(20):     # ~60 ms of back-to-back launches per arm: the clock settles
            _hip.gemm(A, W, o, Mm, N, K, K, K, N, bias=b, tile=v)
    torch.cuda.synchronize()
print("done")
