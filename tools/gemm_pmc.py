"""Runs each GEMM form once per tile kernel (for rocprofv3 --pmc runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
M = N = K = 4096
for form in ("nt", "nn", "tn"):
    a_kc = form[0] == "n"; b_kc = form[1] == "t"
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for tile in (128, 256):
        for _ in range(3):
            _hip.gemm(A, B, out, M, N, K, K, K, N, a_kc=a_kc, b_kc=b_kc, tile=tile)
torch.cuda.synchronize()
