"""Per-tensor error of the HIP path against the fp32 oracle / the reference goldens, beside the bf16 spread that is the
yardstick of the tolerance contract (tests/golden/bf16_spread.npz for the fixtures; the oracle run in plain bf16 for the
smoke configuration).  python tools/tol_probe.py [dact0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]
import numpy as np, torch
from oracle.harness import build_pair, nerr
from climate_learn import _ops
from climate_learn.metrics import Bayesian_TV
from climate_learn.trainer import training_step
if "dact0" in sys.argv:
    _ops._DACT = False
vw = {"total_precipitation_24hr": 1.0}
for seed in (0, 1, 2):
    model, sd, cfg, O, x, y, in_vars, out_vars = build_pair(seed=seed)
    dev = torch.device("cuda:0")
    model = model.to(dev).eval()
    loss = training_step((x, y, in_vars, out_vars), 0, model, dev, vw, Bayesian_TV(aggregate_only=True))
    loss.backward()
    def run(dt):
        s = {k: v.clone().to(dt).requires_grad_() for k, v in sd.items()}
        l = O.training_loss(s, cfg, x.to(dt), y, in_vars, out_vars, "bayesian_tv", vw)
        l.float().backward()
        return float(l), {k: v.grad.float() for k, v in s.items() if v.grad is not None}
    l32, g32 = run(torch.float32)
    l16, g16 = run(torch.bfloat16)
    rows = []
    for n, p in model.named_parameters():
        g = p.grad if p.grad is not None else getattr(p, "_o2g", None)
        if g is None or n not in g32:
            continue
        rows.append((nerr(g, g32[n]), nerr(g16[n], g32[n]), n))
    rows.sort(reverse=True)
    print("seed %d loss hip %.6f oracle %.6f bf16-oracle %.6f" % (seed, float(loss), l32, l16))
    for e, s_, n in rows[:14]:
        print("   %-34s hip %.3e   oracle-in-bf16 %.3e   ratio %.2f" % (n, e, s_, e / max(s_, 1e-12)))
    print("   worst ratio hip / max(2e-2, 1.5 spread): %.2f" % max(e / max(2e-2, 1.5 * s_) for e, s_, _ in rows))
