"""Round 6 diagnostic (library built with -DO2_W4_TRACE: tools/mkvar_gemm.sh w4trace -DO2_W4_TRACE): per-workgroup start / end times
of the 4-wave kernel's K sweep in the Block's grouped weight-gradient launch.  Per round (256 blocks) and XCD (b & 7): the spread of
the cohort's start times and of its end times, in K-tiles of the sweep (duration / 2048) -- L2 holds ~10 K-tiles of a cohort's strips.
   ORBIT2_W4_PACE=0|1 python tools/w4_trace.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
_hip.LIB_PATH = os.environ.get("ORBIT2_TRACE_LIB", os.path.join(ROOT, "orbit-2_amd", "lib", "alt", "w4trace.so"))
T, D = 131072, 3072
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
pad = lambda n: n + 64 if (2 * n) % 8192 == 0 else n
probs = []
for no, ni in ((3 * D, D), (D, D), (4 * D, D), (D, 4 * D)):
    dy, x = r(T, pad(no))[:, :no], r(T, pad(ni))[:, :ni]
    probs.append((dy, x, torch.empty(no, ni, dtype=torch.bfloat16, device="cuda"), no, ni, T, pad(no), pad(ni), ni, dict(a_kc=False, b_kc=False)))
for _ in range(3):
    _hip.gemm_grouped(probs)
torch.cuda.synchronize()
n = 1728
buf = (C.c_ulonglong * (2 * n))()
_hip.lib().orbit2_debug_read_w4_trace(buf, 2 * n)
st = [buf[2 * b] for b in range(n)]
en = [buf[2 * b + 1] for b in range(n)]
t0 = min(st)
ktile = sum(e - s for s, e in zip(st, en)) / n / 2048.0          # ticks (10 ns) per K-tile
print("pace %s: %d workgroups, mean sweep %.1f us = %.3f us per K-tile" % (os.environ.get("ORBIT2_W4_PACE", "1"), n, ktile * 2048 / 100, ktile / 100))
print("round xcd  n   start spread [K-tiles]   end spread [K-tiles]   sweep min/max [us]")
for rd in range((n + 255) // 256):
    for x in range(8):
        bs = [b for b in range(rd * 256, min(n, rd * 256 + 256)) if b & 7 == x]
        s_ = [st[b] for b in bs]; e_ = [en[b] for b in bs]; d_ = [en[b] - st[b] for b in bs]
        print("%5d %3d %3d   %10.1f   %18.1f      %8.1f / %8.1f   first start %.1f us" %
              (rd, x, len(bs), (max(s_) - min(s_)) / ktile, (max(e_) - min(e_)) / ktile, min(d_) / 100, max(d_) / 100, (min(s_) - t0) / 100))

# raw: round 0, per XCD, the 32 sweeps in slot order (slot s = tile id 32 x + s: tm = s % 4 within the group, tn = s // 4)
print("round 0 sweeps [us] by slot, per XCD")
for x in range(8):
    bs = sorted(b for b in range(0, 256) if b & 7 == x)
    print("xcd %d: " % x + " ".join("%6.0f" % ((en[b] - st[b]) / 100.0) for b in bs))
print("round 1 start offsets [us] after the cohort's first start, by slot")
for x in range(8):
    bs = sorted(b for b in range(256, 512) if b & 7 == x)
    m = min(st[b] for b in bs)
    print("xcd %d: " % x + " ".join("%6.1f" % ((st[b] - m) / 100.0) for b in bs))
