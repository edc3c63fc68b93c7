"""Does the LayerNorm backward depend on the state the step leaves the memory system in?  (No: the 0.66 ms launches of the step
profile are simply the batch-16 shape, twice the rows of this probe.)
Times orbit2_layernorm_bwd at [65536, 3072] (a) on one set of buffers over and over, (b) rotating over 24 sets (29 GB: every
call reads buffers last touched 23 calls earlier), (c) rotating, with 1 or 16 MFMA-heavy 3 ms GEMMs in front of every call (sustained power draw, as in the step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
M, D, NS = 65536, 3072, 24
g = torch.randn(D, device="cuda").bfloat16(); b = torch.randn(D, device="cuda").bfloat16()
sets = []
for i in range(NS):
    x = torch.randn(M, D, device="cuda").bfloat16()
    sets.append((x, torch.randn(M, D, device="cuda").bfloat16(), torch.randn(M, D, device="cuda").bfloat16()))
y, mean, rstd = _hip.layernorm_fwd(sets[0][0], g, b)
dg = torch.empty(D, device="cuda", dtype=torch.float32); db = torch.empty(D, device="cuda", dtype=torch.float32)
w = torch.randn(3 * D, D, device="cuda").bfloat16(); o = torch.empty(M, 3 * D, device="cuda", dtype=torch.bfloat16)
def run(rotate, gemm, n=48):
    ev = []
    for i in range(n):
        x, dy, dres = sets[i % NS] if rotate else sets[0]
        for _ in range(gemm): _hip.gemm(x, w, o, M, 3 * D, D, D, D, 3 * D, a_kc=True, b_kc=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _hip.layernorm_bwd(dy, x, g, mean, rstd, dres, dg, db); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ev[8:])
    return v[len(v) // 2]
for name, rot, gm in [("same buffers", False, 0), ("rotating 24 sets", True, 0), ("rotating + 1 GEMM (3 ms) in front", True, 1), ("rotating + 16 GEMMs (45 ms) in front", True, 16),
                      ("same buffers + 16 GEMMs in front", False, 16)]:
    print("%-32s %.3f ms" % (name, run(rot, gm)), flush=True)
