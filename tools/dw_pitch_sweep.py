"""Round 6: row pitch of the K-strided operands of the Block's grouped weight-gradient launch (dY [T, no], X [T, ni], T = 131072).
tools/w4_trace.py showed workgroups whose B strip starts at an odd multiple of 512 B sweeping 3-5 % faster than their
neighbours at even multiples: the strips are served at different rates by the memory side.  Sweep the padding (elements) of the
3072- / 9216-wide operands (the 12288-wide ones carry _ld_pad's 64 already) and time the launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
T, D = 131072, 3072
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
variants = [("as in the step (0 / 0 / 64)", {3072: 0, 9216: 0, 12288: 64}), ("64 / 64 / 64", {3072: 64, 9216: 64, 12288: 64}),
            ("128 / 128 / 64", {3072: 128, 9216: 128, 12288: 64}), ("32 / 32 / 64", {3072: 32, 9216: 32, 12288: 64}),
            ("192 / 192 / 192", {3072: 192, 9216: 192, 12288: 192}), ("64 / 0 / 64", {3072: 64, 9216: 0, 12288: 64}),
            ("256 / 256 / 64", {3072: 256, 9216: 256, 12288: 64}), ("0 / 0 / 0", {3072: 0, 9216: 0, 12288: 0})]
sets = []
for name, pad in variants:
    probs, fl = [], 0.0
    for no, ni in ((3 * D, D), (D, D), (4 * D, D), (D, 4 * D)):
        dy, x = r(T, no + pad[no])[:, :no], r(T, ni + pad[ni])[:, :ni]
        probs.append((dy, x, torch.empty(no, ni, dtype=torch.bfloat16, device="cuda"), no, ni, T, no + pad[no], ni + pad[ni], ni, dict(a_kc=False, b_kc=False)))
        fl += 2.0 * no * ni * T
    sets.append((name, probs))
res = {n: [] for n, _ in sets}
for rnd in range(4):
    for n, probs in sets:
        f = lambda: _hip.gemm_grouped(probs)
        if rnd == 0: f(); f()
        res[n].append(timed(f, 5))
ref = sorted(res[sets[0][0]])[1]
for n, _ in sets:
    m = sorted(res[n])[1]
    print("pad (3072 / 9216 / 12288-wide) %-28s %7.3f ms  %5.0f TFLOP/s  %+5.1f %%   %s" % (n, m, fl / m / 1e9, 100 * (ref / m - 1), " ".join("%.2f" % x for x in res[n])), flush=True)
