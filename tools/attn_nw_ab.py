"""8-wave (256-row workgroups, default) vs 4-wave (flag ORBIT2_ATTN_4WAVES of the *_ex entries, the round-1 geometry) attention kernels: equality of
the results (the per-wave arithmetic is the same: bit-identical) and interleaved timing at the interm_1b / interm_117m shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

def fl(nw):
    return _hip.ATTN_4WAVES if nw == 4 else 0

def t(f, n=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
med = lambda v: sorted(v)[len(v) // 2]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ok = True
for (H, L, d, Bx) in [(24, 8192, 128, B), (16, 4096, 64, B), (2, 256 + 96, 64, 2), (2, 1024 + 32, 128, 1)]:
    qkv = (torch.randn(Bx, L, 3 * H * d, device="cuda") * 0.7).to(torch.bfloat16)
    do = torch.randn(Bx, L, H * d, device="cuda").to(torch.bfloat16)
    for p in (0.0, 0.1):
        res = {}
        for nw in (4, 8):
            out, lse = _hip.attn_fwd(qkv, Bx, L, H, d, p, 11, flags=fl(nw))
            dqkv = _hip.attn_bwd(qkv, out, do, lse, Bx, L, H, d, p, 11, flags=fl(nw))
            res[nw] = (out, lse, dqkv)
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(res[4], res[8]))
        print("H=%d L=%d d=%d p=%.1f: 8-wave == 4-wave bitwise: %s" % (H, L, d, p, same), flush=True)
        ok = ok and same
        if L < 4096:
            continue
        tf, tb = {4: [], 8: []}, {4: [], 8: []}
        for rnd in range(3):
            for nw in (4, 8):
                tf[nw].append(t(lambda: _hip.attn_fwd(qkv, Bx, L, H, d, p, 11, flags=fl(nw))))
                tb[nw].append(t(lambda: _hip.attn_bwd(qkv, res[8][0], do, res[8][1], Bx, L, H, d, p, 11, flags=fl(nw))))
        gf = 4.0 * Bx * H * L * L * d / 1e9
        print("   fwd: 4-wave %7.3f ms %5.0f TF | 8-wave %7.3f ms %5.0f TF (%+.1f %%)   bwd: 4-wave %7.3f ms %5.0f TF | 8-wave %7.3f ms %5.0f TF (%+.1f %%)"
              % (med(tf[4]), gf / med(tf[4]), med(tf[8]), gf / med(tf[8]), 100 * (med(tf[4]) / med(tf[8]) - 1),
                 med(tb[4]), 2 * gf / med(tb[4]), med(tb[8]), 2 * gf / med(tb[8]), 100 * (med(tb[4]) / med(tb[8]) - 1)), flush=True)
print("ALL OK" if ok else "MISMATCH")
