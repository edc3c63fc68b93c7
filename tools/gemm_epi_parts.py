"""fc1-forward GEMM (batch-16 shape, padded output pitch) with its epilogue built up option by option -- bias / + GELU / + saved
GELU' factor / + dropout -- on the 8-phase kernel (tile hint 256) and the 4-wave kernel (260), interleaved rounds in one process:
what each option costs on each kernel.  argv: tokens (default 131072)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip, _ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
D, Hd = 3072, 12288
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
x, w, b = r(T, D), r(Hd, D), r(Hd)
out = _ops._rows(T, Hd, "cuda")
dact = _ops._rows(T, Hd, "cuda", torch.int16)
def t(f, n=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = lambda tile, **kw: _hip.gemm(x, w, out, T, Hd, D, D, D, out.stride(0), tile=tile, **kw)
cases = {
  "none                 ": dict(),
  "bias                 ": dict(bias=b),
  "bias+gelu            ": dict(bias=b, act=1),
  "bias+gelu+dact       ": dict(bias=b, act=1, save_dact=dact),
  "bias+drop            ": dict(bias=b, drop_p=0.1, seed=4),
  "bias+gelu+drop       ": dict(bias=b, act=1, drop_p=0.1, seed=4),
  "bias+gelu+dact+drop  ": dict(bias=b, act=1, save_dact=dact, drop_p=0.1, seed=4),
}
for name, kw in cases.items():
    best = {256: [], 260: []}
    for rnd in range(4):
        for tile in (256, 260):
            if rnd == 0: g(tile, **kw)
            best[tile].append(t(lambda: g(tile, **kw)))
    m6, m7 = sorted(best[256])[2], sorted(best[260])[2]
    print("%s | 8-phase %7.3f ms | 4-wave %7.3f ms" % (name, m6, m7), flush=True)
