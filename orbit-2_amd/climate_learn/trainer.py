"""Training-step glue of the hot path (reference: examples/intermediate_downscaling.py:267-306, 706-753)."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import _ops
from .data.processing.era5_constants import CONSTANTS


def clip_replace_constant(y, yhat, out_variables: List[str]):
    """Clamp the precipitation channel at 0 in place; constant output channels take the ground truth
    (reference :267-278; raises ValueError when 'total_precipitation_24hr' is not an output, like .index())."""
    pi = out_variables.index("total_precipitation_24hr")
    yhat = _ops.ClampChannelFn.apply(yhat, pi)
    for i, name in enumerate(out_variables):
        if name in CONSTANTS:
            yhat[:, i] = y[:, i, : yhat.shape[2], : yhat.shape[3]]
    return yhat


def training_step(batch, batch_idx, net, device, var_weights: Optional[Dict[str, float]], train_loss_metric):
    """forward -> clip -> loss; the target is consumed through its top-left crop (no copy) when larger than
    the prediction (reference :295-296)."""
    x, y, in_variables, out_variables = batch
    x = x.to(device, non_blocking=True)
    y = y.to(device, non_blocking=True)
    yhat = net.forward(x, in_variables, out_variables)
    yhat = clip_replace_constant(y, yhat, out_variables)
    losses = train_loss_metric(yhat, y, var_names=out_variables, var_weights=var_weights)
    return losses if losses.dim() == 0 else losses[-1]


def evaluate_func(batch, stage: str, net, device, loss_metrics, target_transforms):
    """Forward in the caller's mode -> clip -> every metric of `loss_metrics` on (optionally transformed) prediction
    and target; returns {"<stage>/<metric>:<var>": value, ..., "<stage>/<metric>:aggregate": value}
    (reference :321-363).  The target may be larger than the prediction: the metrics read its top-left crop in
    place (the reference slices it, :350-351)."""
    if stage not in ("val", "test"):
        raise RuntimeError("Invalid evaluation stage")
    x, y, in_variables, out_variables = batch
    x = x.to(device, non_blocking=True)
    y = y.to(device, non_blocking=True)
    yhat = net.forward(x, in_variables, out_variables)
    yhat = clip_replace_constant(y, yhat, out_variables)
    loss_dict = {}
    for i, lf in enumerate(loss_metrics):
        yhat_, y_ = yhat, y
        if target_transforms is not None and target_transforms[i] is not None:
            yhat_ = target_transforms[i](yhat_)
            y_ = target_transforms[i](y_)
        losses = lf(yhat_, y_)
        loss_name = getattr(lf, "name", f"loss_{i}")
        if losses.dim() == 0:       # aggregate loss  (the reference's key spells it 'agggregate', :357; kept)
            loss_dict[f"{stage}/{loss_name}:agggregate"] = losses
        else:                       # per channel + aggregate
            for var_name, loss in zip(out_variables, losses):
                loss_dict[f"{stage}/{loss_name}:{var_name}"] = loss
            loss_dict[f"{stage}/{loss_name}:aggregate"] = losses[-1]
    return loss_dict


def validation_step(batch, batch_idx, net, device, val_loss_metrics, val_target_transforms):
    """reference :310-318"""
    return evaluate_func(batch, "val", net, device, val_loss_metrics, val_target_transforms)


def test_step(batch, batch_idx, net, device, test_loss_metrics, test_target_transforms):
    """reference :test stage of evaluate_func (it reads an undefined `self` there, :341; the intended argument is used)"""
    return evaluate_func(batch, "test", net, device, test_loss_metrics, test_target_transforms)
