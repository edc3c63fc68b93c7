"""Forward attention at the interm_1b shape: the generated one-wave-per-SIMD kernel (csrc/attn_fwd_asm.h) against the
compiler-scheduled 8-wave kernel (flag ORBIT2_ATTN_NO_W4), interleaved rounds in one process, q pre-scaled, random data.
usage: attn_w4_ab.py [B] [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, L, d = 24, 8192, 128
torch.manual_seed(0)
qkv = torch.randn(B, L, 3, H * d, device="cuda")
qkv[:, :, 0] *= 1.4426950408889634 / d ** 0.5
qkv = qkv.reshape(B, L, 3 * H * d).to(torch.bfloat16)
PRE = _hip.ATTN_Q_PRESCALED


def t(f, n=5):
    f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


fl = 4.0 * B * H * L * L * d
do = torch.randn(B, L, H * d, device="cuda").to(torch.bfloat16)
for p in (0.1, 0.0):
    o_, l_ = _hip.attn_fwd(qkv, B, L, H, d, p, 7, flags=PRE)
    g_new = _hip.attn_bwd(qkv, o_, do, l_, B, L, H, d, p, 7, flags=PRE)
    g_old = _hip.attn_bwd(qkv, o_, do, l_, B, L, H, d, p, 7, flags=PRE | _hip.ATTN_NO_W4)
    torch.cuda.synchronize()
    gq_n, gq_o = g_new.view(B, L, 3, H * d)[:, :, 0].float(), g_old.view(B, L, 3, H * d)[:, :, 0].float()
    rel = lambda i: float((g_new.view(B, L, 3, H * d)[:, :, i].float() - g_old.view(B, L, 3, H * d)[:, :, i].float()).abs().max()
                          / g_old.view(B, L, 3, H * d)[:, :, i].float().abs().max())
    print("p=%.1f  bwd: max |w4 - old| / max|old|: dq %.3e  dk %.3e  dv %.3e" % (p, rel(0), rel(1), rel(2)), flush=True)
    rb = {"w4": [], "old": []}
    for r in range(R):
        rb["w4"].append(t(lambda: _hip.attn_bwd(qkv, o_, do, l_, B, L, H, d, p, 7, flags=PRE)))
        rb["old"].append(t(lambda: _hip.attn_bwd(qkv, o_, do, l_, B, L, H, d, p, 7, flags=PRE | _hip.ATTN_NO_W4)))
    for k in ("old", "w4"):
        v = sorted(rb[k])
        print("p=%.1f B=%d  bwd %-4s median %7.3f ms (min %7.3f)  %6.0f TFLOP/s algorithmic" % (p, B, k, v[len(v) // 2], v[0], 2 * fl / v[len(v) // 2] / 1e9), flush=True)
for p in (0.1, 0.0):
    o_new, l_new = _hip.attn_fwd(qkv, B, L, H, d, p, 7, flags=PRE)
    o_old, l_old = _hip.attn_fwd(qkv, B, L, H, d, p, 7, flags=PRE | _hip.ATTN_NO_W4)
    torch.cuda.synchronize()
    err = float((o_new.float() - o_old.float()).abs().max() / o_old.float().abs().max())
    lerr = float((l_new - l_old).abs().max())
    print("p=%.1f  max |out_w4 - out_old| / max|out| = %.3e   max |lse diff| = %.3e" % (p, err, lerr), flush=True)
    res = {"w4": [], "old": []}
    for r in range(R):
        res["w4"].append(t(lambda: _hip.attn_fwd(qkv, B, L, H, d, p, 7, flags=PRE)))
        res["old"].append(t(lambda: _hip.attn_fwd(qkv, B, L, H, d, p, 7, flags=PRE | _hip.ATTN_NO_W4)))
    for k in ("old", "w4"):
        v = sorted(res[k])
        med = v[len(v) // 2]
        print("p=%.1f B=%d  %-4s median %7.3f ms (min %7.3f)  %6.0f TFLOP/s" % (p, B, k, med, v[0], fl / med / 1e9), flush=True)
