"""Times the folded variable-aggregation kernels at the interm_1b shape (B x 23 x 128 x 256, D=3072, 24 heads)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

def t(f, n=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for (V, h, w, H, D) in [(23, 128, 256, 24, 3072), (23, 32, 64, 16, 1024), (23, 128, 256, 32, 8192)]:
    x = torch.randn(B, V, h, w, device="cuda")
    stab = torch.randn(H, V, 5, device="cuda") * 0.3
    gtab = torch.randn(V, 5, D, device="cuda") * 0.1
    ntok = B * h * w // 4
    dz = torch.randn(ntok, D, device="cuda").to(torch.bfloat16)
    z, attw = _hip.varagg_fwd(x, stab, gtab, H, D)
    f = t(lambda: _hip.varagg_fwd(x, stab, gtab, H, D))
    b = t(lambda: _hip.varagg_bwd(x, gtab, attw, dz, H, D))
    print("B=%d V=%d grid=%dx%d H=%d D=%d | fwd %7.3f ms | bwd %7.3f ms (incl. zeroing the two tables)" % (B, V, h, w, H, D, f, b), flush=True)
