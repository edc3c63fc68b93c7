import os, sys
import torch
BF = torch.bfloat16
def t(f, n=6):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M, D, hid = 8 * 8192, 3072, 12288
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(BF)
for name, N, K in (("qkv dX", 3 * D, D), ("fc1 dX", hid, D), ("proj dX", D, D)):
    dy, W, Wt, o = r(M, N), r(N, K), r(K, N), torch.empty(M, K, dtype=BF, device="cuda")
    a = t(lambda: torch.matmul(dy, W, out=o))          # NN: W as stored [N, K]
    b = t(lambda: torch.matmul(dy, Wt.t(), out=o))     # NT: transposed copy [K, N]
    fl = 2.0 * M * N * K
    print("%-8s NN %.3f ms %5.0f TF | NT (transposed copy) %.3f ms %5.0f TF" % (name, a, fl / a / 1e9, b, fl / b / 1e9), flush=True)
