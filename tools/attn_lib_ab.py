"""Same-box, same-process A/B of two builds of the attention kernels:
    tools/ab_build.sh <git-rev> <name>;   on the GPU box:  python tools/attn_lib_ab.py orbit-2_amd/lib/alt/<name>.so [B]
loads the working-tree library and the other build side by side (ctypes), checks that forward and backward agree and times
them interleaved (median of 5 rounds of 4 launches) at the interm_1b shape and at d = 64 / d = 256."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

alt = C.CDLL(os.path.abspath(sys.argv[1]))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
BF, F32 = torch.bfloat16, torch.float32
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def fwd(lib, qkv, out, lse, Bx, L, H, d, p):
    assert lib.orbit2_attn_fwd(P(qkv), P(out), P(lse), Bx, L, H, d, C.c_float(p), C.c_uint64(11), S()) == 0
def bwd(lib, qkv, out, do, lse, delta, dqkv, Bx, L, H, d, p):
    assert lib.orbit2_attn_bwd(P(qkv), P(out), P(do), P(lse), P(delta), P(dqkv), Bx, L, H, d, C.c_float(p), C.c_uint64(11), S()) == 0
def t(f, n=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
med = lambda v: sorted(v)[len(v) // 2]
libs = {"tree": _hip.lib(), "alt": alt}
for (H, L, d, Bx) in [(24, 8192, 128, B), (16, 4096, 64, B), (32, 4096, 256, max(1, B // 2)), (2, 1024 + 32, 128, 1)]:
    qkv = (torch.randn(Bx, L, 3 * H * d, device="cuda") * 0.7).to(BF)
    do = torch.randn(Bx, L, H * d, device="cuda").to(BF)
    for p in (0.0, 0.1):
        r = {}
        for k, lib in libs.items():
            out = torch.empty(Bx, L, H * d, dtype=BF, device="cuda"); lse = torch.empty(Bx, H, L, dtype=F32, device="cuda")
            delta = torch.empty(2 * Bx * H * ((L + 63) // 64 * 64 + 64), dtype=F32, device="cuda"); dqkv = torch.empty_like(qkv)   # >= orbit2_attn_bwd_ws_floats
            fwd(lib, qkv, out, lse, Bx, L, H, d, p); bwd(lib, qkv, out, do, lse, delta, dqkv, Bx, L, H, d, p)
            r[k] = (out, lse, delta, dqkv)
        torch.cuda.synchronize()
        eq = [torch.equal(a, b) for a, b in zip(r["tree"], r["alt"])]
        dev = max((a.float() - b.float()).abs().max().item() for a, b in zip(r["tree"], r["alt"]))
        line = "H=%d L=%d d=%d p=%.1f: out/lse/delta/dqkv bitwise equal %s, max |diff| %.2e" % (H, L, d, p, eq, dev)
        if L >= 4096:
            tf, tb = {k: [] for k in libs}, {k: [] for k in libs}
            for rnd in range(5):
                for k, lib in libs.items():
                    o, l, dl, dq = r[k]
                    tf[k].append(t(lambda: fwd(lib, qkv, o, l, Bx, L, H, d, p)))
                    tb[k].append(t(lambda: bwd(lib, qkv, o, do, l, dl, dq, Bx, L, H, d, p)))
            fl = 4.0 * Bx * H * L * L * d / 1e9
            line += "\n   fwd: alt %7.3f ms %5.0f TF | tree %7.3f ms %5.0f TF (%+.1f %%)   bwd: alt %7.3f ms %5.0f TF | tree %7.3f ms %5.0f TF (%+.1f %%)" % (
                med(tf["alt"]), fl / med(tf["alt"]), med(tf["tree"]), fl / med(tf["tree"]), 100 * (med(tf["alt"]) / med(tf["tree"]) - 1),
                med(tb["alt"]), 2 * fl / med(tb["alt"]), med(tb["tree"]), 2 * fl / med(tb["tree"]), 100 * (med(tb["alt"]) / med(tb["tree"]) - 1))
        print(line, flush=True)
