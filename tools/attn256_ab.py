"""d = 256 (interm_10b) attention backward: fused dK+dV pass (default) against the two-pass form (flag ORBIT2_ATTN_SPLIT_DKV),
interleaved timing at the interm_10b shape (32 heads, L = 8192, B = 1) and a shorter one; bitwise comparison of dqkv."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
def t(f, n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
med = lambda v: sorted(v)[len(v) // 2]
for (B, H, L, d) in [(1, 32, 8192, 256), (2, 32, 4096, 256)]:
    qkv = (torch.randn(B, L, 3 * H * d, device="cuda") * 0.7).to(torch.bfloat16)
    do = torch.randn(B, L, H * d, device="cuda").to(torch.bfloat16)
    for p in (0.0, 0.1):
        out, lse = _hip.attn_fwd(qkv, B, L, H, d, p, 11)
        r = {f: _hip.attn_bwd(qkv, out, do, lse, B, L, H, d, p, 11, flags=f) for f in (0, _hip.ATTN_SPLIT_DKV)}
        torch.cuda.synchronize()
        tb = {0: [], _hip.ATTN_SPLIT_DKV: []}
        for rnd in range(5):
            for f in tb:
                tb[f].append(t(lambda: _hip.attn_bwd(qkv, out, do, lse, B, L, H, d, p, 11, flags=f)))
        fl = 8.0 * B * H * L * L * d / 1e9
        print("B=%d H=%d L=%d d=%d p=%.1f: bitwise equal %s | bwd fused %7.3f ms %5.0f TF | split %7.3f ms %5.0f TF (%+.1f %%)"
              % (B, H, L, d, p, torch.equal(r[0], r[_hip.ATTN_SPLIT_DKV]), med(tb[0]), fl / med(tb[0]), med(tb[2]), fl / med(tb[2]),
                 100 * (med(tb[2]) / med(tb[0]) - 1)), flush=True)
