"""Callable loss objects (reference: metrics/metrics.py).  Training-path losses only: mse, bayesian_tv and the
*intended* lat_mse (the reference's LatWeightedMSE passes its arguments positionally into
the wrong slots and rejects the var_names/var_weights kwargs training_step always passes; here it computes
mse(pred, target, var_names, var_weights, aggregate_only, lat_weights[:H_pred]))."""
from typing import Dict, List, Optional

import numpy as np
import torch

from .functional import bayesian_tv, image_gradient, mse
from .utils import MetricsMetaInfo, register


class Metric:
    def __init__(self, aggregate_only: bool = False, metainfo: Optional[MetricsMetaInfo] = None):
        self.aggregate_only = aggregate_only
        self.metainfo = metainfo

    def __call__(self, pred, target):
        raise NotImplementedError()


class LatitudeWeightedMetric(Metric):
    def __init__(self, aggregate_only: bool = False, metainfo: Optional[MetricsMetaInfo] = None):
        super().__init__(aggregate_only, metainfo)
        w = np.cos(np.deg2rad(np.asarray(self.metainfo.lat, dtype=np.float64)))
        self.lat_weights = torch.from_numpy(w / w.mean()).float().view(1, 1, -1, 1)

    def cast_to_device(self, pred):
        self.lat_weights = self.lat_weights.to(device=pred.device)


@register("mse")
class MSE(Metric):
    def __call__(self, pred, target, var_names: Optional[List[str]] = None,
                 var_weights: Optional[Dict[str, float]] = None):
        return mse(pred, target, var_names, var_weights, self.aggregate_only)


@register("bayesian_tv")
class Bayesian_TV(Metric):
    def __call__(self, pred, target, var_names: Optional[List[str]] = None,
                 var_weights: Optional[Dict[str, float]] = None):
        return bayesian_tv(pred, target, var_names, var_weights, self.aggregate_only)


@register("imagegradient")
class IMAGEGRADIENT(Metric):
    def __call__(self, pred, target, var_names: Optional[List[str]] = None,
                 var_weights: Optional[Dict[str, float]] = None):
        return image_gradient(pred, target, var_names, var_weights)


@register("lat_mse")
class LatWeightedMSE(LatitudeWeightedMetric):
    def __call__(self, pred, target, var_names: Optional[List[str]] = None,
                 var_weights: Optional[Dict[str, float]] = None):
        self.cast_to_device(pred)
        return mse(pred, target, var_names, var_weights, self.aggregate_only, self.lat_weights)


def _not_on_path(name):
    class _M(Metric):
        def __call__(self, *a, **k):
            raise NotImplementedError(
                "%s is an evaluation / perceptual metric outside the training hot path of this build" % name)
    _M.__name__ = name.upper()
    return register(name)(_M)


for _n in ("rmse", "pearson", "mean_bias", "mae", "lat_rmse", "lat_acc", "acc", "perceptual"):
    _not_on_path(_n)
