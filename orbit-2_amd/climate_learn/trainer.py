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
