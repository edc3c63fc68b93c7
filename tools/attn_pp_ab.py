"""Software-pipelined ping-pong attention forward (default at >= 256 tokens, d = 64 / 128) vs the plain 8-wave loop
(ORBIT2_ATTN_FWD=plain): bit-equality (same arithmetic order per accumulator) and interleaved timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

def run(mode, f):
    if mode == "plain": os.environ["ORBIT2_ATTN_FWD"] = "plain"
    else: os.environ.pop("ORBIT2_ATTN_FWD", None)
    return f()

def t(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
med = lambda v: sorted(v)[len(v) // 2]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ok = True
for (H, L, d, Bx) in [(24, 8192, 128, B), (16, 4096, 64, B), (16, 512, 64, 8), (2, 256 + 96, 64, 2), (2, 1024 + 32, 128, 1), (3, 256, 128, 2), (2, 320, 128, 1)]:
    qkv = (torch.randn(Bx, L, 3 * H * d, device="cuda") * 0.7).to(torch.bfloat16)
    for p in (0.0, 0.1):
        res = {}
        for mode in ("plain", "pp"):
            res[mode] = run(mode, lambda: _hip.attn_fwd(qkv, Bx, L, H, d, p, 11))
        torch.cuda.synchronize()
        same = torch.equal(res["plain"][0], res["pp"][0]) and torch.equal(res["plain"][1], res["pp"][1])
        fin = bool(torch.isfinite(res["pp"][0].float()).all())
        print("H=%d L=%d d=%d B=%d p=%.1f: ping-pong == plain bitwise: %s (finite %s)" % (H, L, d, Bx, p, same, fin), flush=True)
        ok = ok and same and fin
        if L < 4096:
            continue
        tt = {"plain": [], "pp": []}
        for rnd in range(4):
            for mode in ("plain", "pp"):
                tt[mode].append(run(mode, lambda: t(lambda: _hip.attn_fwd(qkv, Bx, L, H, d, p, 11))))
        fl = 4.0 * Bx * H * L * L * d / 1e9
        print("   fwd: plain 8-wave %7.3f ms %5.0f TF | ping-pong %7.3f ms %5.0f TF (%+.1f %%)"
              % (med(tt["plain"]), fl / med(tt["plain"]), med(tt["pp"]), fl / med(tt["pp"]), 100 * (med(tt["plain"]) / med(tt["pp"]) - 1)), flush=True)
print("ALL OK" if ok else "MISMATCH")
