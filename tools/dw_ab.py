"""Grouped weight-gradient launch of one interm_1b Block (B=4) + two plain forms, for same-box A/B runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
def t(f, n=6):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
T, D = 32768, 3072
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
probs, fl = [], 0.0
for no, ni in ((3 * D, D), (D, D), (4 * D, D), (D, 4 * D)):
    dy, x = r(T, no), r(T, ni)
    probs.append((dy, x, torch.empty(no, ni, dtype=torch.bfloat16, device="cuda"), no, ni, T, no, ni, ni, dict(a_kc=False, b_kc=False)))
    fl += 2.0 * no * ni * T
ms = t(lambda: _hip.gemm_grouped(probs))
print("grouped dW  %7.3f ms %6.0f TF" % (ms, fl / ms / 1e9))
A, B = r(16384, 12288), r(12288, 16384)       # nt and tn on the 128 kernel, 12288 x 12288 x 16384
o = torch.empty(12288, 12288, dtype=torch.bfloat16, device="cuda")
ms = t(lambda: _hip.gemm(A, A, o, 12288, 12288, 16384, 12288, 12288, 12288, a_kc=False, b_kc=False, tile=128), 3)
print("tn128       %7.3f ms %6.0f TF" % (ms, 2.0 * 12288 * 12288 * 16384 / ms / 1e9))
ms = t(lambda: _hip.gemm(B, B, o, 12288, 12288, 16384, 16384, 16384, 12288, tile=128), 3)
print("nt128       %7.3f ms %6.0f TF" % (ms, 2.0 * 12288 * 12288 * 16384 / ms / 1e9))
