import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
M = N = K = 8192
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for tile in (256, 128):
    for _ in range(6):
        _hip.gemm(A, B, out, M, N, K, K, K, N, tile=tile)
torch.cuda.synchronize()
