"""Times orbit2_gemm_bf16 on the interm_1b GEMM shapes (random data), both tile kernels, all three forms."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

def bench(M, N, K, form, tile, iters=10):
    a_kc = form[0] == "n"; b_kc = form[1] == "t"
    A = torch.randn((M, K) if a_kc else (K, M), device="cuda").to(torch.bfloat16)
    B = torch.randn((N, K) if b_kc else (K, N), device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    f = lambda: _hip.gemm(A, B, out, M, N, K, A.shape[1], B.shape[1], N, a_kc=a_kc, b_kc=b_kc, tile=tile)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9

if __name__ == "__main__":
    Mtok = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    D = 3072
    shapes = [("qkv fwd", Mtok, 3 * D, D, "nt"), ("proj fwd", Mtok, D, D, "nt"), ("fc1 fwd", Mtok, 4 * D, D, "nt"),
              ("fc2 fwd", Mtok, D, 4 * D, "nt"), ("qkv dX", Mtok, D, 3 * D, "nn"), ("fc1 dX", Mtok, D, 4 * D, "nn"),
              ("fc2 dX", Mtok, 4 * D, D, "nn"), ("proj dW", D, D, Mtok, "tn"), ("qkv dW", 3 * D, D, Mtok, "tn"),
              ("fc1 dW", 4 * D, D, Mtok, "tn"), ("fc2 dW", D, 4 * D, Mtok, "tn"), ("4096^3", 4096, 4096, 4096, "nt"),
              ("8192^3", 8192, 8192, 8192, "nt")]
    for name, M, N, K, form in shapes:
        r = {t: bench(M, N, K, form, t) for t in (128, 256)}
        print("%-9s %s M=%6d N=%6d K=%6d | 128: %7.3f ms %6.0f TF | 256: %7.3f ms %6.0f TF" %
              (name, form, M, N, K, r[128][0], r[128][1], r[256][0], r[256][1]), flush=True)
