"""Same-process A/B of two builds of the bf16 GEMM (orbit2_gemm_bf16 / _grouped) on a Block's shapes at the interm_1b size:
    python tools/gemm_lib_ab.py orbit-2_amd/lib/alt/<name>.so [--nocheck]
NT forward GEMMs (with their epilogues), NN input gradients, the grouped TN weight gradients; results compared, times
interleaved (median of 5 rounds of 3 launches)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip

alt = C.CDLL(os.path.abspath(sys.argv[1]))
check = "--nocheck" not in sys.argv
libs = {"tree": _hip.lib(), "alt": alt}
BF = torch.bfloat16
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(f, n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
med = lambda v: sorted(v)[len(v) // 2]
M, D = 65536, 3072
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5).to(BF)
cases = []
for name, N, K, kw in [("qkv fwd (bias)", 3 * D, D, dict(bias=True)), ("proj fwd (bias+drop+res)", D, D, dict(bias=True, drop_p=0.1, residual=True)),
                       ("fc1 fwd (bias+gelu+drop)", 4 * D, D, dict(bias=True, act=1, drop_p=0.1, save_pre=True)),
                       ("fc1 fwd kind 1 (+factor)", 4 * D, D, dict(bias=True, act=1, drop_p=0.1, save_dact=True)),
                       ("fc2 fwd (bias+drop+res)", D, 4 * D, dict(bias=True, drop_p=0.1, residual=True))]:
    cases.append((name, "nt", N, K, kw))
for name, N, K in [("dX qkv (NN)", D, 3 * D), ("dX fc1 (NN)", D, 4 * D), ("dX fc2 (NN)", 4 * D, D)]:
    cases.append((name, "nn", N, K, {}))
for name, kind, N, K, kw in cases:
    x = rnd(M, K)
    w = rnd(N, K) if kind == "nt" else rnd(K, N)
    bias = rnd(N) if kw.get("bias") else None
    res = rnd(M, N) if kw.get("residual") else None
    outs = {}
    def call(lib, out, pre):
        a = _hip.GemmArgs()
        k2 = dict(kw); k2.pop("bias", None); k2.pop("residual", None); k2.pop("save_pre", None); k2.pop("save_dact", None)
        _hip._gemm_fill(a, x, w, out, M, N, K, K, K if kind == "nt" else N, N, a_kc=True, b_kc=(kind == "nt"), bias=bias,
                        residual=res, ldr=N if res is not None else 0, save_pre=pre if kw.get("save_pre") else None,
                        save_dact=pre if kw.get("save_dact") else None, seed=5, **k2)
        assert lib.orbit2_gemm_bf16(C.byref(a), S()) == 0
    tm = {k: [] for k in libs}
    for k, lib in libs.items():
        side = torch.empty(M, N, dtype=BF, device="cuda") if kw.get("save_pre") else (torch.empty(M, N, dtype=torch.int16, device="cuda") if kw.get("save_dact") else None)
        outs[k] = (torch.empty(M, N, dtype=BF, device="cuda"), side)
        call(lib, *outs[k])
    torch.cuda.synchronize()
    same = (torch.equal(outs["tree"][0], outs["alt"][0]) and (outs["tree"][1] is None or torch.equal(outs["tree"][1], outs["alt"][1]))) if check else None
    for r in range(5):
        for k, lib in libs.items():
            tm[k].append(t(lambda: call(lib, *outs[k])))
    fl = 2.0 * M * N * K / 1e9
    print("%-28s M=%d N=%5d K=%5d | alt %7.3f ms %5.0f TF | tree %7.3f ms %5.0f TF (%+.1f %%)  equal: %s"
          % (name, M, N, K, med(tm["alt"]), fl / med(tm["alt"]), med(tm["tree"]), fl / med(tm["tree"]), 100 * (med(tm["alt"]) / med(tm["tree"]) - 1), same), flush=True)
    del x, w, outs
# grouped weight gradients of a Block (TN): dW[N,K] = dy[M,N]^T . x[M,K]
probs = []
for N, K in [(3 * D, D), (D, D), (4 * D, D), (D, 4 * D)]:
    probs.append((rnd(M, N), rnd(M, K), N, K))
def grouped(lib, outs):
    arr = (_hip.GemmArgs * 4)()
    for i, (dy, x, N, K) in enumerate(probs):
        _hip._gemm_fill(arr[i], dy, x, outs[i], N, K, M, N, K, K, a_kc=False, b_kc=False)
    assert lib.orbit2_gemm_bf16_grouped(arr, 4, S()) == 0
outs = {k: [torch.empty(N, K, dtype=BF, device="cuda") for (_, _, N, K) in probs] for k in libs}
tm = {k: [] for k in libs}
for k, lib in libs.items(): grouped(lib, outs[k])
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(outs["tree"], outs["alt"])) if check else None
for r in range(5):
    for k, lib in libs.items(): tm[k].append(t(lambda: grouped(lib, outs[k])))
fl = sum(2.0 * M * N * K for (_, _, N, K) in probs) / 1e9
print("%-28s                          | alt %7.3f ms %5.0f TF | tree %7.3f ms %5.0f TF (%+.1f %%)  equal: %s"
      % ("grouped 4 dW (TN)", med(tm["alt"]), fl / med(tm["alt"]), med(tm["tree"]), fl / med(tm["tree"]), 100 * (med(tm["alt"]) / med(tm["tree"]) - 1), same), flush=True)
