import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/orbit-2_amd", "/root/repo/tests"]
import numpy as np, torch
import climate_learn as cl
from climate_learn import _ops
from test_model_gpu import load, VW, nerr, rel_l2, CASES
from climate_learn.metrics import Bayesian_TV
from climate_learn.trainer import clip_replace_constant
G = "/root/repo/tests/golden"
for tag in CASES:
    for dact in (False, True):
        _ops._DACT = dact
        c, z, sd, m = load(G, tag)
        x, y = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["y"]).cuda()
        pred = m(x, c["in_vars"], c["out_vars"])
        yhat = clip_replace_constant(y, pred, c["out_vars"])
        full = Bayesian_TV(aggregate_only=False)(yhat, y, var_names=c["out_vars"], var_weights=VW)
        full[-1].backward()
        rows = []
        for n, p in m.named_parameters():
            k = "g.bayesian_tv." + n
            if k in z.files:
                rows.append((nerr(p.grad, z[k]), rel_l2(p.grad, z[k]), n))
        rows.sort(reverse=True)
        print(tag, "dact" if dact else "pre ", "pred nerr %.4f" % nerr(pred, z["pred"]), " worst:", [(round(a, 4), round(b, 4), n) for a, b, n in rows[:3]], " max l2 %.4f" % max(b for _, b, _ in rows), flush=True)
