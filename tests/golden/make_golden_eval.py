#!/usr/bin/env python3
"""Golden vectors for the evaluation metrics (rmse / pearson / mean_bias, reference metrics/functional.py:236-324),
produced by the REFERENCE's own functions with the same import recipe as make_golden.py.  Build container only
(needs /root/reference); writes tests/golden/eval_metrics.npz (numeric inputs and outputs only)."""
import importlib
import os

import numpy as np
import torch

from make_golden import OUT, install_shims, t2n


def main():
    install_shims()
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
    fn = importlib.import_module("climate_learn.metrics.functional")
    g = torch.Generator().manual_seed(123)
    pred = torch.randn(3, 3, 24, 40, generator=g) * 1.7 + 0.3
    target = 0.6 * pred + torch.randn(3, 3, 24, 40, generator=g)
    lat = np.linspace(-88.0, 88.0, 24)
    wl = np.cos(np.deg2rad(lat))
    wl = torch.from_numpy(wl / wl.mean()).view(1, 1, -1, 1).float()
    out = {"pred": t2n(pred), "target": t2n(target), "lat": lat,
           "rmse": t2n(fn.rmse(pred, target, False)), "rmse.agg": t2n(fn.rmse(pred, target, True)),
           "lat_rmse": t2n(fn.rmse(pred, target, False, wl)),
           "pearson": t2n(fn.pearson(pred, target, False)), "pearson.agg": t2n(fn.pearson(pred, target, True)),
           "mean_bias": t2n(fn.mean_bias(pred, target, False)), "mean_bias.agg": t2n(fn.mean_bias(pred, target, True))}
    clim = torch.randn(3, 24, 40, generator=g) * 0.5
    out.update({"clim": t2n(clim), "mae": t2n(fn.mae(pred, target, False)), "lat_mae": t2n(fn.mae(pred, target, False, wl)),
                "lat_acc": t2n(fn.acc(pred, target, clim.unsqueeze(0), False, wl)),
                "lat_acc.agg": t2n(fn.acc(pred, target, clim.unsqueeze(0), True, wl)),
                "acc_unit_weights": t2n(fn.acc(pred, target, clim.unsqueeze(0), False, torch.ones_like(wl)))})
    np.savez_compressed(os.path.join(OUT, "eval_metrics.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
