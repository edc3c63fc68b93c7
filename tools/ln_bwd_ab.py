"""LayerNorm backward of the Block (131072 x 3072, with the residual stream's gradient fused) on two builds of the library,
interleaved: lib/alt/ln0.so (-DO2_LN_FMA=0: the expressions of rounds 2-5) against the product; bytes moved = 4 tensors."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip
libs = [("ln0", C.CDLL(os.path.join(ROOT, "orbit-2_amd", "lib", "alt", "ln0.so"))), ("product", C.CDLL(_hip.LIB_PATH))]
M, D = 131072, 3072
x = torch.randn(M, D, device="cuda").bfloat16(); dy = torch.randn(M, D, device="cuda").bfloat16(); dres = torch.randn(M, D, device="cuda").bfloat16()
g = torch.randn(D, device="cuda").bfloat16(); b = torch.randn(D, device="cuda").bfloat16()
_hip._lib = libs[1][1]
y, mean, rstd = _hip.layernorm_fwd(x, g, b)
def t(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res, outs = {n: [] for n, _ in libs}, {}
for rnd in range(5):
    for n, lib in libs:
        _hip._lib = lib
        dg = torch.empty(D, device="cuda", dtype=torch.bfloat16); db = torch.empty(D, device="cuda", dtype=torch.bfloat16)
        f = lambda: _hip.layernorm_bwd(dy, x, g, mean, rstd, dres, dg, db)
        if rnd == 0: outs[n] = (f().clone(), dg.clone(), db.clone())
        res[n].append(t(f))
for n, _ in libs:
    m = sorted(res[n])[2]
    print("%-8s %.1f us  %.2f TB/s (4 x %d x %d x 2 B)   rounds %s" % (n, 1e3 * m, 4 * M * D * 2 / m / 1e9, M, D, " ".join("%.1f" % (1e3 * v) for v in res[n])))
a, b_ = outs["ln0"], outs["product"]
print("dx max |diff| %.3e (bf16 ulps of the output scale), dgamma equal %s, dbeta equal %s" % (float((a[0].float() - b_[0].float()).abs().max()), torch.equal(a[1], b_[1]), torch.equal(a[2], b_[2])))
ref = ((dy.float() * g.float()) - (dy.float() * g.float()).mean(1, keepdim=True) - ((x.float() - mean[:, None]) * rstd[:, None]) * ((dy.float() * g.float()) * ((x.float() - mean[:, None]) * rstd[:, None])).mean(1, keepdim=True))[:4096] * rstd[:4096, None] + dres[:4096].float()
for n in outs:
    print("%-8s dx vs fp32 on 4096 rows: max err %.3e" % (n, float((outs[n][0][:4096].float() - ref).abs().max())))
