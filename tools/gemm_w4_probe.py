"""Timing of the 4-wave kernel's schedule variants / timing-only probes (library built by tools/mkvar_gemm.sh w4probe
-DO2_W4_PROBES: tile hints 260 + V, csrc/gemm_w4_asm.h) against the 8-phase kernel (256), interleaved rounds in one process.
Probes with dropped loads / reads / barriers compute wrong results by construction: only their time is read."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
_hip.LIB_PATH = os.path.join(ROOT, "orbit-2_amd", "lib", "alt", "w4probe.so")
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
def t(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
variants = [256] + [260 + int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5,6,7".split(","))]
D = 3072
for name, Mm, N, K in (("qkv", 65536, 3 * D, D), ("fc2", 65536, D, 4 * D), ("8192^3", 8192, 8192, 8192), ("longK", 4096, 4096, 65536)):
    A, W, b = r(Mm, K), r(N, K), r(N)
    o = torch.empty(Mm, N, dtype=torch.bfloat16, device="cuda")
    best = {v: [] for v in variants}
    for v in variants:
        _hip.gemm(A, W, o, Mm, N, K, K, K, N, bias=b, tile=v)
    for rnd in range(4):
        for v in variants:
            best[v].append(t(lambda: _hip.gemm(A, W, o, Mm, N, K, K, K, N, bias=b, tile=v)))
    f = 2.0 * Mm * N * K / 1e9
    med = {v: sorted(best[v])[len(best[v]) // 2] for v in variants}
    print("%-7s M=%6d N=%6d K=%6d | " % (name, Mm, N, K) + " | ".join("%d: %6.3f ms %5.0f TF" % (v, med[v], f / med[v]) for v in variants), flush=True)
