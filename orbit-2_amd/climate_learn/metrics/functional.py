"""Functional losses of the hot path on the fused HIP kernels (reference: metrics/functional.py:117-202).

mse / bayesian_tv: one forward kernel (error map + latitude / variable weights + reduction) and one backward
kernel, instead of ~15-25 elementwise ATen launches.  Argument order matches the reference."""
from typing import Dict, List, Optional

import torch

from .. import _ops


def _chan_weights(pred, var_names, var_weights):
    if var_names is None:
        return None
    assert len(var_names) == pred.shape[1], "Number of variable names must match channel dimension"
    w = [float((var_weights or {}).get(v, 1.0)) for v in var_names]
    return torch.tensor(w, dtype=torch.float32, device=pred.device)


def _lat(lat_weights, pred):
    if lat_weights is None:
        return None
    w = lat_weights.reshape(-1).to(device=pred.device, dtype=torch.float32)
    return w[: pred.shape[2]].contiguous()      # prediction may be a top-left crop of the target grid


def _fused(pred, target, var_names, var_weights, aggregate_only, lat_weights, kind):
    if isinstance(pred, torch.distributions.Normal):
        pred = pred.loc
    out = _ops.LossFn.apply(pred, target.float(), _lat(lat_weights, pred), _chan_weights(pred, var_names, var_weights),
                            kind)
    return out[-1] if aggregate_only else out


def mse(pred, target, var_names: Optional[List[str]] = None, var_weights: Optional[Dict[str, float]] = None,
        aggregate_only: bool = False, lat_weights=None):
    return _fused(pred, target, var_names, var_weights, aggregate_only, lat_weights, 0)


def bayesian_tv(pred, target, var_names: Optional[List[str]] = None, var_weights: Optional[Dict[str, float]] = None,
                aggregate_only: bool = False, lat_weights=None):
    return _fused(pred, target, var_names, var_weights, aggregate_only, lat_weights, 1)


def image_gradient(pred, target, var_names: Optional[List[str]] = None, var_weights: Optional[Dict[str, float]] = None,
                   aggregate_only: bool = False, lat_weights=None):
    """mean((pred-target)^2 * w_var) + 0.1 * mean(|grad(target) - grad(pred)|) * mean(w_var)  (reference :59-114;
    forward-difference gradients as torchmetrics.functional.image.image_gradients defines them).  Scalar."""
    return _fused(pred, target, var_names, var_weights, True, None, 2)
