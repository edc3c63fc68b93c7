"""Times the pre-scaled d = 128 attention forward of SEVERAL builds of the library in one process, interleaved rounds
(schedule variants / timing-only ablations of the generated kernel: tools/mkvar_af.sh):
   python tools/attn_fwd_multi_ab.py lib1.so lib2.so ... [--batch B] [--rounds R]
The first library is the reference of the printed differences.  interm_1b shape (24 heads, L = 8192, d = 128), p = 0.1 and 0."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
paths = [a for a in sys.argv[1:] if a.endswith(".so")]
B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 4
R = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 5
flags = int(sys.argv[sys.argv.index("--flags") + 1]) if "--flags" in sys.argv else 4
libs = [(os.path.basename(p), C.CDLL(os.path.abspath(p))) for p in paths]
BF, F32 = torch.bfloat16, torch.float32
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
H, L, d = 24, 8192, 128
torch.manual_seed(0)
qkv = torch.randn(B, L, 3, H * d, device="cuda")
qkv[:, :, 0] *= 1.4426950408889634 / d ** 0.5
qkv = qkv.reshape(B, L, 3 * H * d).to(BF)
def t(f, n=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
med = lambda v: sorted(v)[len(v) // 2]
fl = 4.0 * B * H * L * L * d
BWD = "--bwd" in sys.argv
do = torch.randn(B, L, H * d, device="cuda").to(BF)
for p in (0.1, 0.0):
    st = {}
    for name, lib in libs:
        st[name] = (torch.empty(B, L, H * d, dtype=BF, device="cuda"), torch.empty(B, H, L, dtype=F32, device="cuda"))
    def fwd(lib, s):
        assert lib.orbit2_attn_fwd_ex(P(qkv), P(s[0]), P(s[1]), B, L, H, d, C.c_float(p), C.c_uint64(11), flags, S()) == 0
    if BWD:      # time the backward (statistics + dQ + dK/dV) instead; the printed difference is over dqkv
        for name, lib in libs:
            fwd(lib, st[name])
            lib.orbit2_attn_bwd_ws_floats.restype = C.c_int64
            ws = torch.empty(int(lib.orbit2_attn_bwd_ws_floats(B, L, H)), dtype=F32, device="cuda")
            st[name] = (torch.empty_like(qkv), st[name][1], st[name][0], ws)
        def fwd(lib, s):
            assert lib.orbit2_attn_bwd_ex(P(qkv), P(s[2]), P(do), P(s[1]), P(s[3]), P(s[0]), B, L, H, d, C.c_float(p), C.c_uint64(11), flags, S()) == 0
    for name, lib in libs:
        fwd(lib, st[name]); fwd(lib, st[name])
    torch.cuda.synchronize()
    ref = st[libs[0][0]]
    tf = {n: [] for n, _ in libs}
    for r in range(R):
        for name, lib in libs:
            tf[name].append(t(lambda: fwd(lib, st[name])))
    print("p = %.1f  B = %d" % (p, B))
    for name, _ in libs:
        diff = float((st[name][0].float() - ref[0].float()).abs().max())
        print("   %-22s fwd %7.3f ms (min %7.3f)  %5.0f TFLOP/s  (%+5.1f %% vs first)   max |out - first| %.2e"
              % (name, med(tf[name]), min(tf[name]), fl / med(tf[name]) / 1e9, 100 * (med(tf[libs[0][0]]) / med(tf[name]) - 1), diff), flush=True)
