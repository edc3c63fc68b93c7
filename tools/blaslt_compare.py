"""How the library GEMM (torch.matmul -> hipBLASLt / rocBLAS) does on the Block's plain shapes, beside orbit2_gemm_bf16.
Reference measurement only: the product path never calls it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
BF = torch.bfloat16
def t(f, n=6):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M, D, hid = B * 8192, 3072, 12288
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(BF)
print("shape                                  |  orbit2 ms   TF | library ms   TF")
# NT: y[M,N] = x[M,K] . W[N,K]^T
for name, N, K in (("qkv fwd  NT", 3 * D, D), ("fc1 fwd  NT", hid, D), ("fc1 dX   NT (K=4D)", D, hid), ("proj     NT", D, D)):
    x, w, o = r(M, K), r(N, K), torch.empty(M, N, dtype=BF, device="cuda")
    a = t(lambda: _hip.gemm(x, w, o, M, N, K, K, K, N))
    b = t(lambda: torch.matmul(x, w.t(), out=o))
    fl = 2.0 * M * N * K
    print("%-38s | %8.3f %5.0f | %8.3f %5.0f" % (name, a, fl / a / 1e9, b, fl / b / 1e9), flush=True)
# TN (weight gradient): dW[N,K] = dy[M,N]^T . x[M,K]
for name, N, K in (("qkv dW   TN", 3 * D, D), ("fc1 dW   TN", hid, D), ("fc2 dW   TN", D, hid), ("proj dW  TN", D, D)):
    dy, x, o = r(M, N), r(M, K), torch.empty(N, K, dtype=BF, device="cuda")
    a = t(lambda: _hip.gemm(dy, x, o, N, K, M, N, K, K, a_kc=False, b_kc=False))
    b = t(lambda: torch.matmul(dy.t(), x, out=o))
    fl = 2.0 * M * N * K
    print("%-38s | %8.3f %5.0f | %8.3f %5.0f" % (name, a, fl / a / 1e9, b, fl / b / 1e9), flush=True)
