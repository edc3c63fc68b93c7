import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
D = 3072; Mtok = 16384
def run(M, N, K, form, tile):
    a_kc = form[0] == "n"; b_kc = form[1] == "t"
    A = torch.randn((M, K) if a_kc else (K, M), device="cuda").to(torch.bfloat16)
    B = torch.randn((N, K) if b_kc else (K, N), device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        _hip.gemm(A, B, out, M, N, K, A.shape[1], B.shape[1], N, a_kc=a_kc, b_kc=b_kc, tile=tile)
run(Mtok, 4 * D, D, "nt", 128)     # fc1 fwd
run(4 * D, D, Mtok, "tn", 128)     # fc1 dW
run(Mtok, 4 * D, D, "nn", 128)     # fc2 dX
torch.cuda.synchronize()
