"""Callable loss objects (reference: metrics/metrics.py).  Training-path losses only: mse, bayesian_tv, perceptual and the
*intended* lat_mse (the reference's LatWeightedMSE passes its arguments positionally into
the wrong slots and rejects the var_names/var_weights kwargs training_step always passes; here it computes
mse(pred, target, var_names, var_weights, aggregate_only, lat_weights[:H_pred]))."""
from typing import Dict, List, Optional

import numpy as np
import torch

from .functional import acc, bayesian_tv, image_gradient, mae, mean_bias, mse, pearson, rmse
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


@register("perceptual")
class PERCEPTUAL(Metric):
    """L1 + 0.5 * mean LPIPS-VGG16 (reference: metrics.py:119-187, functional.py:17-33).  Same constructor as the
    reference (device, model, aggregate_only, metainfo).  The reference's __call__ takes (pred, target) only, while
    training_step always passes var_names= / var_weights=: they are accepted and ignored here.
    LPIPS weights: a state dict file named by $ORBIT2_LPIPS_WEIGHTS (lpips / torchvision key names).  The reference
    loads lpips' PRETRAINED VGG16 (`lpips.LPIPS(net='vgg')`); training against anything else is a different loss, so a
    missing file is an error.  Seeded random stand-in weights (synthetic throughput runs, parity tests against the
    restated graph) must be asked for explicitly: `synthetic_weights=True` or ORBIT2_LPIPS_SYNTHETIC=1.
    Export the real ones on a machine that has the packages:
        import lpips, torch; torch.save(lpips.LPIPS(net='vgg').state_dict(), 'lpips_vgg.pt')"""

    graph_capturable = True       # the backward keeps the upstream scalar on the device (lpips_hip._PerceptualFn.backward)

    def __init__(self, device, model, aggregate_only: bool = False, metainfo: Optional[MetricsMetaInfo] = None,
                 synthetic_weights: Optional[bool] = None):
        import os
        from .lpips_hip import LPIPSVGG16
        super().__init__(aggregate_only, metainfo)
        self.model = model
        path = os.environ.get("ORBIT2_LPIPS_WEIGHTS")
        if synthetic_weights is None:
            synthetic_weights = os.environ.get("ORBIT2_LPIPS_SYNTHETIC", "0") == "1"
        state = None
        if path:
            state = torch.load(path, map_location="cpu", weights_only=True)
        elif not synthetic_weights:
            raise RuntimeError(
                "perceptual loss: no LPIPS-VGG16 weights.  Set $ORBIT2_LPIPS_WEIGHTS to a state-dict file exported from "
                "lpips.LPIPS(net='vgg') (or torchvision vgg16 + lpips lins); seeded random stand-in weights are only "
                "used when asked for explicitly (synthetic_weights=True / ORBIT2_LPIPS_SYNTHETIC=1): a run that "
                "optimises 0.5*LPIPS against a random network is not the reference's loss.")
        self.loss_fn = LPIPSVGG16(device, state)

    def __call__(self, pred, target, var_names: Optional[List[str]] = None,
                 var_weights: Optional[Dict[str, float]] = None):
        return self.loss_fn.perceptual(pred, target)


@register("perceptual_lat_mse")
class PerceptualLatMSE(PERCEPTUAL, LatitudeWeightedMetric):
    """BASELINE.json configs[4] names a "hybrid perceptual + lat-weighted MSE loss" (SURVEY 8d-5): the sum of the reference's
    `perceptual` (metrics.py:119-187: L1 + 0.5 * mean LPIPS-VGG16) and its intended `lat_mse` (metrics.py:295-316: the
    variable- and latitude-weighted MSE, aggregate over channels).  The reference has no object for the sum; this one takes
    the perceptual loss's constructor (device, model, ...) and training_step's keyword arguments."""

    def __init__(self, device, model, aggregate_only: bool = False, metainfo: Optional[MetricsMetaInfo] = None,
                 synthetic_weights: Optional[bool] = None):
        PERCEPTUAL.__init__(self, device, model, aggregate_only, metainfo, synthetic_weights)
        w = np.cos(np.deg2rad(np.asarray(self.metainfo.lat, dtype=np.float64)))
        self.lat_weights = torch.from_numpy(w / w.mean()).float().view(1, 1, -1, 1)

    def __call__(self, pred, target, var_names: Optional[List[str]] = None,
                 var_weights: Optional[Dict[str, float]] = None):
        self.cast_to_device(pred)
        per = self.loss_fn.perceptual(pred, target)
        lat = mse(pred, target, var_names, var_weights, True, self.lat_weights)
        return per + lat


@register("rmse")
class RMSE(Metric):
    """reference metrics.py RMSE: unweighted root mean squared error (validation / test metric of the downscaling task)"""
    def __call__(self, pred, target, mask=None):
        return rmse(pred, target, self.aggregate_only, mask=mask)


@register("pearson")
class Pearson(Metric):
    def __call__(self, pred, target):
        return pearson(pred, target, self.aggregate_only)


@register("mean_bias")
class MeanBias(Metric):
    def __call__(self, pred, target):
        return mean_bias(pred, target, self.aggregate_only)


@register("lat_rmse")
class LatWeightedRMSE(LatitudeWeightedMetric):
    def __call__(self, pred, target, mask=None):
        self.cast_to_device(pred)
        return rmse(pred, target, self.aggregate_only, self.lat_weights, mask)


@register("mae")
class MAE(Metric):
    def __call__(self, pred, target):
        return mae(pred, target, self.aggregate_only)


class ClimatologyBasedMetric(Metric):
    """metrics that subtract the split's climatology (reference metrics.py:78-97): metainfo.climatology is [C,H,W]"""

    def __init__(self, aggregate_only: bool = False, metainfo: Optional[MetricsMetaInfo] = None):
        Metric.__init__(self, aggregate_only, metainfo)
        self.climatology = self.metainfo.climatology.unsqueeze(0)


@register("acc")
class ACC(ClimatologyBasedMetric):
    def __call__(self, pred, target, mask=None):
        return acc(pred, target, self.climatology, self.aggregate_only, None, mask)


@register("lat_acc")
class LatWeightedACC(LatitudeWeightedMetric, ClimatologyBasedMetric):
    def __init__(self, aggregate_only: bool = False, metainfo: Optional[MetricsMetaInfo] = None):
        LatitudeWeightedMetric.__init__(self, aggregate_only, metainfo)
        ClimatologyBasedMetric.__init__(self, aggregate_only, metainfo)

    def __call__(self, pred, target, mask=None):
        self.cast_to_device(pred)
        return acc(pred, target, self.climatology, self.aggregate_only, self.lat_weights, mask)
