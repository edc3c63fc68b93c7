"""Functional losses of the hot path on the fused HIP kernels (reference: metrics/functional.py:117-202).

mse / bayesian_tv: one forward kernel (error map + latitude / variable weights + reduction) and one backward
kernel, instead of ~15-25 elementwise ATen launches.  Argument order matches the reference."""
from typing import Dict, List, Optional

import torch

from .. import _ops


_CW_CACHE = {}


def _chan_weights(pred, var_names, var_weights):
    if var_names is None:
        return None
    assert len(var_names) == pred.shape[1], "Number of variable names must match channel dimension"
    w = tuple(float((var_weights or {}).get(v, 1.0)) for v in var_names)
    key = (w, str(pred.device))
    t = _CW_CACHE.get(key)       # cached: no host-to-device copy inside a step (a captured hipGraph forbids one)
    if t is None:
        t = _CW_CACHE[key] = torch.tensor(w, dtype=torch.float32, device=pred.device)
    return t


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


# ---- evaluation metrics (reference :236-324); one reduction kernel, the [B,C,6] -> [C+1] algebra on the host ----------
def _moments(pred, target, lat_weights=None, climatology=None):
    from .. import _hip
    if isinstance(pred, torch.distributions.Normal):
        pred = pred.loc
    clim = None
    if climatology is not None:
        clim = climatology.detach().to(device=pred.device, dtype=torch.float32)
        clim = clim.reshape(-1, *clim.shape[-2:])[:, : pred.shape[2], : pred.shape[3]].contiguous()
    return _hip.eval_moments(pred.detach().float().contiguous(), target.detach().float().contiguous(),
                             _lat(lat_weights, pred), clim), pred.shape[2] * pred.shape[3]


def _with_aggregate(per_channel, aggregate_only):
    agg = per_channel.mean()
    return agg if aggregate_only else torch.cat((per_channel, agg.unsqueeze(0)))


def rmse(pred, target, aggregate_only: bool = False, lat_weights=None, mask=None):
    """sqrt(mean_hw((pred-target)^2 * w_lat)) per (b, c), mean over b, then over c (reference :236-255)."""
    if mask is not None:
        raise NotImplementedError("masked rmse is not on the downscaling path")
    m, n = _moments(pred, target, lat_weights)
    return _with_aggregate((m[..., 5] / n).sqrt().mean(0).float(), aggregate_only)


def pearson(pred, target, aggregate_only: bool = False):
    """cosine similarity of the mean-removed, channel-wise flattened [C, B*H*W] fields (reference :294-308)."""
    m, n = _moments(pred, target)
    s = m.sum(0)                                                   # [C,6] over the batch
    N = n * pred.shape[0]
    cov = s[:, 4] - s[:, 0] * s[:, 1] / N
    vp = (s[:, 2] - s[:, 0] ** 2 / N).clamp_min(0).sqrt().clamp_min(1e-8)      # F.cosine_similarity's eps
    vt = (s[:, 3] - s[:, 1] ** 2 / N).clamp_min(0).sqrt().clamp_min(1e-8)
    return _with_aggregate((cov / (vp * vt)).float(), aggregate_only)


def mean_bias(pred, target, aggregate_only: bool = False):
    """mean(target) - mean(pred) per channel (reference :311-324)."""
    m, n = _moments(pred, target)
    s = m.sum(0)
    return _with_aggregate(((s[:, 1] - s[:, 0]) / (n * pred.shape[0])).float(), aggregate_only)


def mae(pred, target, aggregate_only: bool = False, lat_weights=None):
    """mean |pred - target| (x latitude weight) per channel and over everything (reference :219-232; with equal
    channel sizes the overall mean is the mean of the channel means)."""
    m, n = _moments(pred, target, lat_weights)
    return _with_aggregate((m[..., 6].sum(0) / (n * pred.shape[0])).float(), aggregate_only)


def acc(pred, target, climatology, aggregate_only: bool = False, lat_weights=None, mask=None):
    """anomaly correlation (reference :259-291): anomalies w.r.t. the climatology, each channel centred by its
    unweighted mean over (B,H,W), latitude-weighted covariance / sqrt(variances).  The reference's `mask` argument has
    no effect there (its masked sums are overwritten by the unmasked ones, :282-284) and is ignored here; without
    latitude weights the reference raises (None * tensor) -- unit weights are used instead."""
    m, n = _moments(pred, target, lat_weights, climatology)
    s = m.sum(0)                                        # [C,12] over the batch
    N = n * pred.shape[0]
    ma, mb = s[:, 0] / N, s[:, 1] / N
    if lat_weights is None:
        sw = torch.full_like(ma, float(N))
    else:
        sw = _lat(lat_weights, pred).double().sum() * pred.shape[3] * pred.shape[0]
    cov = s[:, 9] - ma * s[:, 8] - mb * s[:, 7] + ma * mb * sw
    va = s[:, 10] - 2 * ma * s[:, 7] + ma * ma * sw
    vb = s[:, 11] - 2 * mb * s[:, 8] + mb * mb * sw
    return _with_aggregate((cov / (va * vb).sqrt()).float(), aggregate_only)
