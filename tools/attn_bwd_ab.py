"""A/B of attention-backward variants selected by the flags argument of orbit2_attn_bwd_ex (include/orbit2_hip.h):
    python tools/attn_bwd_ab.py flags=2 [--batch B]        (2 = ORBIT2_ATTN_SPLIT_DKV, 1 = ORBIT2_ATTN_4WAVES)
runs the default build and the build with the given switches interleaved on the interm_1b shape (and a ragged and a d = 64
shape for equality), prints the largest deviation of dQ/dK/dV between the two and the median times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

sw = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a and not a.startswith("--"))
B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 4

FLAGS = int(sw.get("flags", "2"))

def t(f, n=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
med = lambda v: sorted(v)[len(v) // 2]
for (H, L, d, Bx) in [(24, 8192, 128, B), (2, 1024 + 32, 128, 1), (3, 512, 128, 2), (2, 256 + 96, 64, 2)]:
    qkv = (torch.randn(Bx, L, 3 * H * d, device="cuda") * 0.7).to(torch.bfloat16)
    do = torch.randn(Bx, L, H * d, device="cuda").to(torch.bfloat16)
    for p in (0.0, 0.1):
        out, lse = _hip.attn_fwd(qkv, Bx, L, H, d, p, 11)
        r0 = _hip.attn_bwd(qkv, out, do, lse, Bx, L, H, d, p, 11)
        r1 = _hip.attn_bwd(qkv, out, do, lse, Bx, L, H, d, p, 11, flags=FLAGS)
        torch.cuda.synchronize()
        dev = (r0.float() - r1.float()).abs().max().item()
        print("H=%d L=%d d=%d p=%.1f: bitwise equal %s, max |diff| %.3e (max |ref| %.3e)"
              % (H, L, d, p, torch.equal(r0, r1), dev, r0.float().abs().max().item()), flush=True)
        if L < 4096:
            continue
        tb = {False: [], True: []}
        for rnd in range(3):
            for on in (False, True):
                tb[on].append(t(lambda: _hip.attn_bwd(qkv, out, do, lse, Bx, L, H, d, p, 11, flags=FLAGS if on else 0)))
        fl = 8.0 * Bx * H * L * L * d / 1e9
        print("   bwd: default %7.3f ms %5.0f TF | %s %7.3f ms %5.0f TF (%+.1f %%)"
              % (med(tb[False]), fl / med(tb[False]), " ".join("%s=%s" % kv for kv in sw.items()), med(tb[True]),
                 fl / med(tb[True]), 100 * (med(tb[False]) / med(tb[True]) - 1)), flush=True)
