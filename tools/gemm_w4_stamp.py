"""Reads the per-segment cycle sums of the stamped 4-wave loop (library built by tools/mkvar_gemm.sh w4stamp -DO2_W4_STAMP; tile
hint 261).  Segments of one K-tile: top .. bar1 (Y reads, first B pieces) | wait + bar1 | bar1 .. bar2 (pieces) | wait + bar2 |
bar2 .. end (X reads, last A pieces); then the per-tile anatomy (first tile of the first 64 workgroups: every CU in phase)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
_hip.LIB_PATH = os.path.join(ROOT, "orbit-2_amd", "lib", "alt", "w4stamp.so")
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
D, T = 3072, 65536
cases = (("NT qkv", T, 3 * D, D, True, True), ("NT fc2", T, D, 4 * D, True, True), ("NN dXqkv", T, D, 3 * D, True, False),
         ("NN dXfc2", T, 4 * D, D, True, False), ("TN dWfc1", 4 * D, D, T, False, False), ("TN dWqkv", 3 * D, D, T, False, False))
for name, M, N, K, a_kc, b_kc in cases:
    A = r(M, K) if a_kc else r(K, M)
    W = r(N, K) if b_kc else r(K, N)
    lda, ldb = (K if a_kc else M), (K if b_kc else N)
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        _hip.gemm(A, W, o, M, N, K, lda, ldb, N, a_kc=a_kc, b_kc=b_kc, tile=261)
    torch.cuda.synchronize()
    buf = (C.c_uint * (64 * 4 * 16))()
    _hip.lib().orbit2_debug_read_w4(buf, 64 * 4 * 16)
    rows = [[buf[(wg * 4 + w) * 16 + k] for k in range(16)] for wg in range(64) for w in range(4)]
    nk = rows[0][5]
    avg = [sum(x[k] for x in rows) / len(rows) / nk for k in range(5)]
    print("%-9s nk=%4d | cycles per K-tile: top..bar1 %5.0f | bar1 wait %5.0f | pieces %5.0f | bar2 wait %5.0f | X reads %5.0f | sum %5.0f (ideal 2048)"
          % (name, nk, *avg, sum(avg)), flush=True)
    tsv = [sum(x[8 + k] for x in rows) / len(rows) for k in range(7)]
    loop = sum(avg) * nk
    print("          per tile (cycles): setup %6.0f | asm prologue (+ stamp overhead) %6.0f | loop %7.0f | C stage 0 %6.0f | rows 0 %6.0f | C stage 1 %6.0f | rows 1 %6.0f | store drain %6.0f | total %7.0f"
          % (tsv[0], tsv[1] - tsv[0] - loop, loop, tsv[2] - tsv[1], tsv[3] - tsv[2], tsv[4] - tsv[3], tsv[5] - tsv[4], tsv[6] - tsv[5], tsv[6]), flush=True)
