"""Reads the per-segment cycle sums of the stamped 4-wave loop (library built by tools/mkvar_gemm.sh w4stamp -DO2_W4_STAMP;
tile hints 268 = base schedule, 269 = early-B-release schedule).  Segments of one K-tile: top .. bar1 (Y reads [+ B pieces]) |
wait + bar1 | bar1 .. bar2 (LDS-DMA pieces) | wait + bar2 | bar2 .. end (X reads)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
_hip.LIB_PATH = os.path.join(ROOT, "orbit-2_amd", "lib", "alt", "w4stamp.so")
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
D = 3072
for name, Mm, N, K in (("qkv", 65536, 3 * D, D), ("fc2", 65536, D, 4 * D), ("longK", 4096, 4096, 65536)):
    A, W, b = r(Mm, K), r(N, K), r(N)
    o = torch.empty(Mm, N, dtype=torch.bfloat16, device="cuda")
    for v in (268, 269):
        for _ in range(3):
            _hip.gemm(A, W, o, Mm, N, K, K, K, N, bias=b, tile=v)
        torch.cuda.synchronize()
        buf = (C.c_uint * (64 * 4 * 16))()
        _hip.lib().orbit2_debug_read_w4(buf, 64 * 4 * 16)
        rows = [[buf[(wg * 4 + w) * 16 + k] for k in range(16)] for wg in range(64) for w in range(4)]
        nk = rows[0][5]
        avg = [sum(x[k] for x in rows) / len(rows) / nk for k in range(5)]
        perw = [[sum(rows[wg * 4 + w][k] for wg in range(64)) / 64 / nk for k in range(5)] for w in range(4)]
        print("%-6s v%d nk=%d | cycles per K-tile: top..bar1 %5.0f | bar1 wait %5.0f | pieces %5.0f | bar2 wait %5.0f | X reads %5.0f | sum %5.0f (ideal 2048)"
              % (name, v, nk, *avg, sum(avg)), flush=True)
        tsv = [sum(x[8 + k] for x in rows) / len(rows) for k in range(7)]
        loop = sum(avg) * nk
        print("        per tile (cycles): setup %6.0f | asm prologue %6.0f | loop %7.0f | C stage 0 %6.0f | rows 0 %6.0f | C stage 1 %6.0f | rows 1 %6.0f | store drain %6.0f | total %7.0f"
              % (tsv[0], tsv[1] - tsv[0] - loop, loop, tsv[2] - tsv[1], tsv[3] - tsv[2], tsv[4] - tsv[3], tsv[5] - tsv[4], tsv[6] - tsv[5], tsv[6]), flush=True)
