"""Round 6, verdict item 2: does the traffic beyond L2 cost the grouped weight-gradient launch its clock?
Arms = builds of the library (tools/mkvar_w4.sh): the product loop, and TIMING-ONLY ablations whose K walk wraps over the first
n K-tiles (`kwrap=n`: same MFMAs, LDS reads and LDS-DMA pieces; the panels come from L2 after the first pass).
  python tools/dw_traffic_ab.py base.so kwrap8.so kwrap2.so          interleaved timing: the Block's four dW products at batch 16
  python tools/dw_traffic_ab.py --pmc                                 workload of one rocprofv3 --pmc pass (library = $ORBIT2_HIP_LIB)
Record: profiles/r06_dw_traffic_clock.txt (tools/dw_traffic.sh)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
T, D = 131072, 3072
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)


def problems():
    probs, fl = [], 0.0
    pad = lambda n: n + 64 if (2 * n) % 8192 == 0 else n        # the step's row pitches (climate_learn/_ops.py:_ld_pad)
    for no, ni in ((3 * D, D), (D, D), (4 * D, D), (D, 4 * D)):
        dy, x = r(T, pad(no))[:, :no], r(T, pad(ni))[:, :ni]
        probs.append((dy, x, torch.empty(no, ni, dtype=torch.bfloat16, device="cuda"), no, ni, T, pad(no), pad(ni), ni, dict(a_kc=False, b_kc=False)))
        fl += 2.0 * no * ni * T
    return probs, fl


def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if "--pmc" in sys.argv:
    probs, fl = problems()
    for _ in range(14): _hip.gemm_grouped(probs)         # ~250 ms of back-to-back launches: the clock settles; the summary keeps the last 8
    torch.cuda.synchronize()
    print("done")
    sys.exit(0)

paths = [a for a in sys.argv[1:] if a.endswith(".so")]
SINGLE = "--no-single" not in sys.argv
libs = [(os.path.basename(p).replace(".so", ""), C.CDLL(os.path.abspath(p))) for p in paths]
probs, fl = problems()
res = {n: [] for n, _ in libs}
for rnd in range(5):
    for n, lib in libs:
        _hip._lib = lib
        f = lambda: _hip.gemm_grouped(probs)
        if rnd == 0: f(); f()
        res[n].append(timed(f, 6))
ref = sorted(res[libs[0][0]])[2]
for n, _ in libs:
    m = sorted(res[n])[2]
    print("grouped dW (4 products, %d tokens) %-10s median %7.3f ms  %5.0f TFLOP/s  %+5.1f %% vs %s   rounds: %s" %
          (T, n, m, fl / m / 1e9, 100 * (ref / m - 1), libs[0][0], " ".join("%.3f" % x for x in res[n])), flush=True)
# the NN / NT single forms (the kernel's other operand forms) under the same arms
cases = (("NT qkv", T, 3 * D, D, True, True), ("NN dXqkv", T, D, 3 * D, True, False), ("NN dXfc1", T, D, 4 * D, True, False)) if SINGLE else ()
for name, M, N, K, a_kc, b_kc in cases:
    A = r(M, K) if a_kc else r(K, M)
    W = r(N, K) if b_kc else r(K, N)
    lda, ldb = (K if a_kc else M), (K if b_kc else N)
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    rs = {n: [] for n, _ in libs}
    for rnd in range(4):
        for n, lib in libs:
            _hip._lib = lib
            f = lambda: _hip.gemm(A, W, o, M, N, K, lda, ldb, N, a_kc=a_kc, b_kc=b_kc, tile=260)
            if rnd == 0: f()
            rs[n].append(timed(f, 5))
    ref = sorted(rs[libs[0][0]])[1]
    print("%-9s " % name + " | ".join("%s %6.3f ms %5.0f TF %+5.1f %%" % (n, sorted(rs[n])[1], 2.0 * M * N * K / sorted(rs[n])[1] / 1e9,
                                                                         100 * (ref / sorted(rs[n])[1] - 1)) for n, _ in libs), flush=True)
