"""Round 6, VERDICT r5 item 1(b), with the kernels that exist: does a second resident workgroup hide a heavy epilogue behind its
partner's main loop, and what does that form pay in the main loop?  The Block's heavy-epilogue GEMMs at batch 16 (fc1 forward =
kind 1: bias + GELU + saved factor + dropout; proj / fc2 forward = kind 2: bias + dropout + row scale + residual; fc2 input gradient =
kind 3: x saved factor) and the SAME products with a bias-only / plain epilogue, on
  * the 128 x 128-tile kernel (tile hint 128: 4 waves, TWO workgroups per CU, compiler-scheduled: one workgroup's epilogue runs
    under the other's main loop -- the occupancy form the verdict asks to measure, at the tile size this kernel has),
  * the 8-phase 256 x 256 kernel (256: 8 waves = two per SIMD, but ONE workgroup: both run the epilogue together),
  * the 4-wave 256 x 256 kernel (260: one wave per SIMD, compile-time epilogue kinds).
heavy - lean = what the epilogue costs each form; lean = what the main loop + store path of each form is worth."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import torch
from climate_learn import _hip, _ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
D, Hd = 3072, 12288
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
x = _ops._rows(T, D, "cuda"); x.copy_(r(T, D))
hm = _ops._rows(T, Hd, "cuda"); hm.copy_(r(T, Hd))
res = r(T, D)
rs = torch.full((T // 8192,), 1.0 / 0.9, device="cuda")
w_proj, w_fc1, w_fc2 = r(D, D), r(Hd, D), r(D, Hd)
b_d, b_h = r(D), r(Hd)
out_d = torch.empty(T, D, dtype=torch.bfloat16, device="cuda")
out_h = _ops._rows(T, Hd, "cuda")
dact = _ops._rows(T, Hd, "cuda", torch.int16)
dact.copy_((torch.rand(T, Hd, device="cuda") * 16384).to(torch.int16))
ldx = x.stride(0)
def t(f, n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cases = [
  ("fc1 fwd  (kind 1)", lambda tile: _hip.gemm(x, w_fc1, out_h, T, Hd, D, ldx, D, out_h.stride(0), bias=b_h, act=1, save_dact=dact, drop_p=0.1, seed=4, tile=tile),
                        lambda tile: _hip.gemm(x, w_fc1, out_h, T, Hd, D, ldx, D, out_h.stride(0), bias=b_h, tile=tile), 2.0 * T * Hd * D),
  ("proj fwd (kind 2)", lambda tile: _hip.gemm(x, w_proj, out_d, T, D, D, ldx, D, D, bias=b_d, drop_p=0.1, seed=3, rowscale=rs, rows_per_scale=8192, residual=res, ldr=D, tile=tile),
                        lambda tile: _hip.gemm(x, w_proj, out_d, T, D, D, ldx, D, D, bias=b_d, tile=tile), 2.0 * T * D * D),
  ("fc2 fwd  (kind 2)", lambda tile: _hip.gemm(hm, w_fc2, out_d, T, D, Hd, hm.stride(0), Hd, D, bias=b_d, drop_p=0.1, seed=5, rowscale=rs, rows_per_scale=8192, residual=res, ldr=D, tile=tile),
                        lambda tile: _hip.gemm(hm, w_fc2, out_d, T, D, Hd, hm.stride(0), Hd, D, bias=b_d, tile=tile), 2.0 * T * D * Hd),
  ("fc2 dX   (kind 3)", lambda tile: _hip.gemm(x, w_fc2, out_h, T, Hd, D, ldx, Hd, out_h.stride(0), a_kc=True, b_kc=False, mul=dact, tile=tile),
                        lambda tile: _hip.gemm(x, w_fc2, out_h, T, Hd, D, ldx, Hd, out_h.stride(0), a_kc=True, b_kc=False, tile=tile), 2.0 * T * Hd * D),
]
names = {128: "128-tile, 2 workgroups / CU", 256: "8-phase, 8 waves", 260: "4-wave"}
print("# tokens %d; median of 4 interleaved rounds of 3 launches; TF = TFLOP/s of the heavy form" % T)
for name, heavy, lean, fl in cases:
    th, tl = {k: [] for k in names}, {k: [] for k in names}
    for rnd in range(4):
        for tile in names:
            if rnd == 0: heavy(tile); lean(tile)
            th[tile].append(t(lambda: heavy(tile)))
            tl[tile].append(t(lambda: lean(tile)))
    line = name
    for tile in names:
        h, l = sorted(th[tile])[2], sorted(tl[tile])[2]
        line += " | %s: heavy %6.3f ms (%4.0f TF) lean %6.3f ms  epilogue costs %+5.1f %%" % (names[tile], h, fl / h / 1e9, l, 100 * (h / l - 1))
    print(line, flush=True)
