"""Would a padded row pitch of the fc2 weight (W2 [3072, 12288]: 24 KiB rows) pay on the 8-phase kernel?  The two GEMMs that read it
with their real epilogues at batch 16, W2 contiguous against W2 with 64 elements of padding per row, interleaved rounds:
  fc2 forward  (NT: A = hidden [T, 12288] padded pitch, B = W2 K-contiguous; bias + dropout + DropPath + residual)
  fc2 input gradient (NN: A = dY [T, 3072], B = W2 K-strided; dropout mask x GELU'(pre) epilogue, output with padded pitch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
BF = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(BF)
def t(f, n=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
T, D, HID, L = 131072, 3072, 12288, 8192
hm = r(T, HID + 64)[:, :HID]
pre = r(T, HID + 64)[:, :HID]
dy, res, b2 = r(T, D), r(T, D), r(D)
dp = torch.full((T // L,), 1.0 / 0.9, device="cuda")
w2c = r(D, HID)
w2p = torch.empty(D, HID + 64, dtype=BF, device="cuda")[:, :HID]
w2p.copy_(w2c)
oD = torch.empty(T, D, dtype=BF, device="cuda")
oH = torch.empty(T, HID + 64, dtype=BF, device="cuda")[:, :HID]
res_t = {}
for rnd in range(3):
    for tag, w in (("contig", w2c), ("padded", w2p)):
        f1 = lambda: _hip.gemm(hm, w, oD, T, D, HID, hm.stride(0), w.stride(0), D, bias=b2, drop_p=0.1, seed=7, rowscale=dp, rows_per_scale=L, residual=res, ldr=D)
        f2 = lambda: _hip.gemm(dy, w, oH, T, HID, D, D, w.stride(0), oH.stride(0), a_kc=True, b_kc=False, drop_p=0.1, seed=6, dgelu_pre=pre)
        if rnd == 0: f1(); f2()
        res_t.setdefault(("fc2 fwd", tag), []).append(t(f1))
        res_t.setdefault(("fc2 dX ", tag), []).append(t(f2))
fl = 2.0 * T * D * HID / 1e9
for k, v in res_t.items():
    m = sorted(v)[1]
    print("%s W2 %s: %7.3f ms %5.0f TF" % (k[0], k[1], m, fl / m), flush=True)
