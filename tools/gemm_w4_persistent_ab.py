"""The Block's GEMMs at the batch-16 shapes with their real epilogues on the 4-wave kernel: persistent launch (tile hint 260, 256
workgroups walking the tiles, next tile's first K-tiles touched into L2 during the epilogue) against one workgroup per tile (263),
interleaved rounds in one process.  argv: tokens (default 131072)
NOTE: the persistent launch and hint 263 existed only in the build this was measured with (profiles/r04_gemm_w4_persistent_ab.txt:
-0.8 % over the Block's eight GEMMs); the committed kernel launches one workgroup per tile."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip, _ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
D, Hd = 3072, 12288
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
x = r(T, D); hm = _ops._rows(T, Hd, "cuda"); hm.copy_(r(T, Hd)); res = r(T, D); qkv = r(T, 3 * D)
rs = torch.full((T // 8192,), 1.0 / 0.9, device="cuda")
w_qkv, w_proj, w_fc1, w_fc2 = r(3 * D, D), r(D, D), r(Hd, D), r(D, Hd)
b_q, b_d, b_h = r(3 * D), r(D), r(Hd)
out_q = torch.empty(T, 3 * D, dtype=torch.bfloat16, device="cuda")
out_d = torch.empty(T, D, dtype=torch.bfloat16, device="cuda")
out_h = _ops._rows(T, Hd, "cuda"); dact = _ops._rows(T, Hd, "cuda", torch.int16)
def t(f, n=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cases = {
  "qkv fwd  bias+colscale      ": (lambda tile: _hip.gemm(x, w_qkv, out_q, T, 3 * D, D, D, D, 3 * D, bias=b_q, colscale=(D, 0.1275), tile=tile), 2.0 * T * 3 * D * D),
  "proj fwd drop+rowscale+res  ": (lambda tile: _hip.gemm(x, w_proj, out_d, T, D, D, D, D, D, bias=b_d, drop_p=0.1, seed=3, rowscale=rs, rows_per_scale=8192, residual=res, ldr=D, tile=tile), 2.0 * T * D * D),
  "fc1 fwd  gelu+dact+drop     ": (lambda tile: _hip.gemm(x, w_fc1, out_h, T, Hd, D, D, D, out_h.stride(0), bias=b_h, act=1, save_dact=dact, drop_p=0.1, seed=4, tile=tile), 2.0 * T * Hd * D),
  "fc2 fwd  drop+rowscale+res  ": (lambda tile: _hip.gemm(hm, w_fc2, out_d, T, D, Hd, hm.stride(0), Hd, D, bias=b_d, drop_p=0.1, seed=5, rowscale=rs, rows_per_scale=8192, residual=res, ldr=D, tile=tile), 2.0 * T * D * Hd),
  "fc2 dX   x factor (NN)      ": (lambda tile: _hip.gemm(x, w_fc2, out_h, T, Hd, D, D, Hd, out_h.stride(0), a_kc=True, b_kc=False, mul=dact, tile=tile), 2.0 * T * Hd * D),
  "fc1 dX   (NN)               ": (lambda tile: _hip.gemm(hm, w_fc1, out_d, T, D, Hd, hm.stride(0), D, D, a_kc=True, b_kc=False, tile=tile), 2.0 * T * D * Hd),
  "proj dX  (NN)               ": (lambda tile: _hip.gemm(x, w_proj, out_d, T, D, D, D, D, D, a_kc=True, b_kc=False, tile=tile), 2.0 * T * D * D),
  "qkv dX   (NN)               ": (lambda tile: _hip.gemm(qkv, w_qkv, out_d, T, D, 3 * D, 3 * D, D, D, a_kc=True, b_kc=False, tile=tile), 2.0 * T * 3 * D * D),
}
tot = {260: 0.0, 263: 0.0}
for name, (f, fl) in cases.items():
    best = {263: [], 260: []}
    for rnd in range(5):
        for tile in best:
            if rnd == 0: f(tile)
            best[tile].append(t(lambda: f(tile)))
    m = {k: sorted(v)[2] for k, v in best.items()}
    for k in tot: tot[k] += m[k]
    print("%s | one tile per workgroup %7.3f ms %5.0f TF | persistent %7.3f ms %5.0f TF | %+.1f %%" %
          (name, m[263], fl / m[263] / 1e9, m[260], fl / m[260] / 1e9, 100 * (m[263] / m[260] - 1)), flush=True)
print("sum: %.3f ms -> %.3f ms" % (tot[263], tot[260]))
