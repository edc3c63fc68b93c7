"""The single-round GEMMs of interm_117m at batch 8 (4096 tokens, N = 1024: 256 tiles of 128x128): 2-deep ring (tile hint 128)
against the 4-deep ring (129), interleaved rounds in one process, many launches per timing (the kernels run 20-50 us).
NOTE: hint 129 existed only in the build this was measured with (profiles/r04_gemm_single_round_ab.txt: the 4-deep ring LOST 10-15 %,
DESIGN 6c); on the committed library it is the 128-tile kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
M = 4096
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
def t(f, n=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K, b_kc in (("proj fwd NT", 1024, 1024, True), ("fc2 fwd NT", 1024, 4096, True), ("dX proj NN", 1024, 1024, False),
                         ("dX qkv NN", 1024, 3072, False), ("dX fc1 NN", 1024, 4096, False), ("qkv fwd NT", 3072, 1024, True)):
    A, B = r(M, K), (r(N, K) if b_kc else r(K, N))
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    f = lambda tile: _hip.gemm(A, B, out, M, N, K, K, K if b_kc else N, N, a_kc=True, b_kc=b_kc, tile=tile)
    res = {128: [], 129: [], 0: []}
    for rnd in range(5):
        for tile in res:
            if rnd == 0: f(tile)
            res[tile].append(t(lambda: f(tile)))
    m = {k: sorted(v)[2] for k, v in res.items()}
    fl = 2.0 * M * N * K
    print("%-12s N=%d K=%d | 2-deep %6.1f us %5.0f TF | 4-deep %6.1f us %5.0f TF | auto %6.1f us" %
          (name, N, K, m[128], fl / m[128] / 1e6, m[129], fl / m[129] / 1e6, m[0]), flush=True)
