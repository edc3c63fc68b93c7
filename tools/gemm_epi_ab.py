"""The Block's three forward GEMMs with their REAL epilogues and row pitches (proj: bias + dropout + DropPath row scale + residual;
fc1: bias + GELU + save_dact + dropout, padded output pitch; fc2: the padded hidden tensor as A, bias + dropout + row scale +
residual), and fc2's input gradient with the multiply-by-factor epilogue: 8-phase kernel (tile hint 256) against the 4-wave kernel
with its runtime epilogue (262) and with the compile-time epilogue kinds (260), interleaved rounds in one process.
argv: tokens (default 131072 = batch 16)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip, _ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
D, Hd = 3072, 12288
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
x = r(T, D)
hm = _ops._rows(T, Hd, "cuda"); hm.copy_(r(T, Hd))
res = r(T, D)
rs = torch.full((T // 8192,), 1.0 / 0.9, device="cuda")
w_proj, w_fc1, w_fc2 = r(D, D), r(Hd, D), r(D, Hd)
b_d, b_h = r(D), r(Hd)
out_d = torch.empty(T, D, dtype=torch.bfloat16, device="cuda")
out_h = _ops._rows(T, Hd, "cuda")
dact = _ops._rows(T, Hd, "cuda", torch.int16)
def t(f, n=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cases = {
  "proj fwd  bias+drop+rowscale+res": (lambda tile: _hip.gemm(x, w_proj, out_d, T, D, D, D, D, D, bias=b_d, drop_p=0.1, seed=3, rowscale=rs, rows_per_scale=8192, residual=res, ldr=D, tile=tile), 2.0 * T * D * D),
  "fc1 fwd   bias+gelu+dact+drop   ": (lambda tile: _hip.gemm(x, w_fc1, out_h, T, Hd, D, D, D, out_h.stride(0), bias=b_h, act=1, save_dact=dact, drop_p=0.1, seed=4, tile=tile), 2.0 * T * Hd * D),
  "fc2 fwd   bias+drop+rowscale+res": (lambda tile: _hip.gemm(hm, w_fc2, out_d, T, D, Hd, hm.stride(0), Hd, D, bias=b_d, drop_p=0.1, seed=5, rowscale=rs, rows_per_scale=8192, residual=res, ldr=D, tile=tile), 2.0 * T * D * Hd),
  "fc2 dX    x factor (NN)         ": (lambda tile: _hip.gemm(x, w_fc2, out_h, T, Hd, D, D, Hd, out_h.stride(0), a_kc=True, b_kc=False, mul=dact, tile=tile), 2.0 * T * Hd * D),
  "fc1 shape bias only             ": (lambda tile: _hip.gemm(x, w_fc1, out_h, T, Hd, D, D, D, out_h.stride(0), bias=b_h, tile=tile), 2.0 * T * Hd * D),
}
for name, (f, fl) in cases.items():
    best = {256: [], 262: [], 260: []}
    for rnd in range(4):
        for tile in best:
            if rnd == 0: f(tile)
            best[tile].append(t(lambda: f(tile)))
    m = {k: sorted(v)[2] for k, v in best.items()}
    print("%s | 8-phase %7.3f ms %5.0f TF | 4-wave runtime epilogue %7.3f ms %5.0f TF | 4-wave %7.3f ms %5.0f TF | %+.1f %% vs 8-phase" %
          (name, m[256], fl / m[256] / 1e9, m[262], fl / m[262] / 1e9, m[260], fl / m[260] / 1e9, 100 * (m[256] / m[260] - 1)), flush=True)
