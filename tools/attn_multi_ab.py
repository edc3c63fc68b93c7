"""Times attention forward / backward of SEVERAL builds of the library in one process, interleaved (compiler-schedule variants
built with -DO2_FV=n / -DO2_QV=n into orbit-2_amd/lib/alt/):  python tools/attn_multi_ab.py lib1.so lib2.so ... [--batch B]
The first library is the reference for the bitwise comparison.  interm_1b shape (24 heads, L = 8192, d = 128), p = 0.1 and 0."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
paths = [a for a in sys.argv[1:] if a.endswith(".so")]
B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 4
libs = [(os.path.basename(p), C.CDLL(os.path.abspath(p))) for p in paths]
BF, F32 = torch.bfloat16, torch.float32
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
H, L, d = 24, 8192, 128
if "--shape" in sys.argv:        # --shape B H L d   (e.g. the interm_10b shape: 1 32 8192 256)
    i = sys.argv.index("--shape")
    B, H, L, d = (int(v) for v in sys.argv[i + 1:i + 5])
qkv = (torch.randn(B, L, 3 * H * d, device="cuda") * 0.7).to(BF)
do = torch.randn(B, L, H * d, device="cuda").to(BF)
def t(f, n=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
med = lambda v: sorted(v)[len(v) // 2]
for p in (0.1, 0.0):
    st = {}
    for name, lib in libs:
        out = torch.empty(B, L, H * d, dtype=BF, device="cuda"); lse = torch.empty(B, H, L, dtype=F32, device="cuda")
        delta = torch.empty(2 * B * H * ((L + 63) // 64 * 64 + 64), dtype=F32, device="cuda"); dq = torch.empty_like(qkv)   # >= orbit2_attn_bwd_ws_floats
        st[name] = (out, lse, delta, dq)
    def fwd(lib, s): assert lib.orbit2_attn_fwd(P(qkv), P(s[0]), P(s[1]), B, L, H, d, C.c_float(p), C.c_uint64(11), S()) == 0
    def bwd(lib, s): assert lib.orbit2_attn_bwd(P(qkv), P(s[0]), P(do), P(s[1]), P(s[2]), P(s[3]), B, L, H, d, C.c_float(p), C.c_uint64(11), S()) == 0
    for name, lib in libs:
        fwd(lib, st[name]); bwd(lib, st[name])
    torch.cuda.synchronize()
    ref = st[libs[0][0]]
    tf, tb = {n: [] for n, _ in libs}, {n: [] for n, _ in libs}
    for r in range(5):
        for name, lib in libs:
            tf[name].append(t(lambda: fwd(lib, st[name]))); tb[name].append(t(lambda: bwd(lib, st[name])))
    print("p = %.1f" % p)
    for name, _ in libs:
        same = all(torch.equal(a, b) for a, b in zip(ref, st[name]))
        print("   %-14s fwd %7.3f ms (%+5.1f %%)   bwd %7.3f ms (%+5.1f %%)   bitwise == first: %s"
              % (name, med(tf[name]), 100 * (med(tf[libs[0][0]]) / med(tf[name]) - 1), med(tb[name]),
                 100 * (med(tb[libs[0][0]]) / med(tb[name]) - 1), same), flush=True)
