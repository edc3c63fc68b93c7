"""Loader / factory API of the hot path (reference: utils/loaders.py): same function names, argument meaning,
return tuples and error messages, building the HIP-backed objects."""
import warnings
from functools import partial
from typing import Any, Callable, Dict, Iterable, Optional, Union

import torch
import torch.nn as nn

from ..metrics import METRICS_REGISTRY, MetricsMetaInfo
from ..models.hub import MODEL_REGISTRY, Res_Slim_ViT
from ..models.lr_scheduler import LinearWarmupCosineAnnealingLR
from ..transforms import TRANSFORMS_REGISTRY
from .fused_attn import FusedAttn


def _issue(what, name):
    return ("%s is not an implemented %s. If you think it should be, please raise an issue" % (name, what))


def load_architecture(task, data_module, architecture, default_vars, superres_mag=4, cnn_ratio=4, patch_size=2,
                      embed_dim=256, depth=6, decoder_depth=1, num_heads=4, mlp_ratio=4, drop_path=0.1, drop_rate=0.1,
                      tensor_par_size=1, tensor_par_group=None, FusedAttn_option=FusedAttn.HIP):
    in_vars, out_vars = data_module.get_data_variables()
    in_shape, out_shape = data_module.get_data_dims()
    if task != "downscaling" or architecture != "res_slimvit":
        raise NotImplementedError(
            f"{architecture} is not an implemented architecture for the {task} task in the MI355X hot-path build "
            "(only downscaling/res_slimvit is in scope).")
    in_channels, in_h, in_w = in_shape[1:]
    out_channels = out_shape[1]
    return Res_Slim_ViT(default_vars, (in_h, in_w), in_channels, out_channels, superres_mag=superres_mag, history=1,
                        patch_size=patch_size, cnn_ratio=cnn_ratio, learn_pos_emb=True, embed_dim=embed_dim,
                        depth=depth, decoder_depth=decoder_depth, num_heads=num_heads, mlp_ratio=mlp_ratio,
                        drop_path=drop_path, drop_rate=drop_rate, tensor_par_size=tensor_par_size,
                        tensor_par_group=tensor_par_group, FusedAttn_option=FusedAttn_option)


def load_loss(device, model, loss_name, aggregate_only, metainfo):
    cls = METRICS_REGISTRY.get(loss_name)
    if cls is None:
        raise NotImplementedError(_issue("loss", loss_name))
    if loss_name in ("perceptual", "perceptual_lat_mse"):      # reference loaders.py:439-441: this loss takes (device, model)
        return cls(device, model, aggregate_only=aggregate_only, metainfo=metainfo)
    return cls(aggregate_only=aggregate_only, metainfo=metainfo)


def load_transform(transform_name, data_module):
    cls = TRANSFORMS_REGISTRY.get(transform_name)
    if cls is None:
        raise NotImplementedError(_issue("transform", transform_name))
    return cls(data_module)


def load_optimizer(net: torch.nn.Module, optim: str, optim_kwargs: Dict[str, Any] = {}):
    """'adamw' -> the fused HIP AdamW (torch.optim.AdamW math); other names are not on the hot path."""
    from ..optim import HipAdamW
    if len(list(net.parameters())) == 0:
        warnings.warn("Net has no trainable parameters, setting optimizer to `None`")
        return None
    if optim.lower() != "adamw":
        raise NotImplementedError(_issue("optimizer", optim))
    engine = net if hasattr(net, "opt_segments") else None          # either data-parallel engine (dist/dp_engine.py, dist/fsdp_engine.py)
    return HipAdamW(net.parameters(), engine=engine, **optim_kwargs)


def load_lr_scheduler(sched: str, optimizer, sched_kwargs: Dict[str, Any] = {}):
    if optimizer is None:
        warnings.warn("Optimizer is `None`, setting LR scheduler to `None` too")
        return None
    if sched == "constant":
        return torch.optim.lr_scheduler.ConstantLR(optimizer, **sched_kwargs)
    if sched == "linear":
        return torch.optim.lr_scheduler.LinearLR(optimizer, **sched_kwargs)
    if sched == "exponential":
        return torch.optim.lr_scheduler.ExponentialLR(optimizer, **sched_kwargs)
    if sched == "linear-warmup-cosine-annealing":
        return LinearWarmupCosineAnnealingLR(optimizer, **sched_kwargs)
    raise NotImplementedError(_issue("learning rate scheduler", sched))


def _climatology(data_module, split):
    clim = data_module.get_climatology(split=split)
    if clim is None:
        raise RuntimeError("Climatology has not yet been set.")
    if isinstance(clim, dict):
        clim = torch.stack(tuple(clim.values()))
    return clim


def load_model_module(device, data_module, task: str, architecture: Optional[str] = None,
                      model: Optional[Union[str, nn.Module]] = None, model_kwargs: Optional[Dict[str, Any]] = None,
                      optim=None, optim_kwargs=None, sched=None, sched_kwargs=None,
                      train_loss: Optional[Union[str, Callable]] = None, val_loss: Optional[Iterable] = None,
                      test_loss: Optional[Iterable] = None, train_target_transform=None, val_target_transform=None,
                      test_target_transform=None):
    lat, lon = data_module.get_lat_lon()
    if lat is None and lon is None:
        raise RuntimeError("Data module has not been set up yet.")
    if architecture is None and model is None:
        raise RuntimeError("Please specify 'architecture' or 'model'")
    if architecture and model is None:
        model = load_architecture(task, data_module, architecture, **(model_kwargs or {}))
    elif isinstance(model, str):
        raise RuntimeError(f"{model} is not an implemented model.")
    elif not isinstance(model, nn.Module):
        raise TypeError("'model' must be str or nn.Module")

    in_vars, out_vars = data_module.get_data_variables()

    def metainfo(split):
        return MetricsMetaInfo(in_vars, out_vars, lat, lon, _climatology(data_module, split))

    if isinstance(train_loss, str):
        train_loss = load_loss(device, model, train_loss, True, metainfo("train"))
    elif not isinstance(train_loss, Callable):
        raise TypeError("'train_loss' must be str or Callable")

    def load_many(names, split):
        out = []
        for n in names or []:
            out.append(load_loss(device, model, n, False, metainfo(split)) if isinstance(n, str) else n)
        return out

    def load_tf(specs):
        if specs is None:
            return None
        out = []
        for s in specs:
            out.append(load_transform(s, data_module) if isinstance(s, str) else s)
        return out

    val_losses, test_losses = load_many(val_loss, "val"), load_many(test_loss, "test")
    train_transform = load_transform(train_target_transform, data_module) \
        if isinstance(train_target_transform, str) else train_target_transform
    return (model, train_loss, val_losses, test_losses, train_transform, load_tf(val_target_transform),
            load_tf(test_target_transform))


load_forecasting_module = partial(load_model_module, task="forecasting", train_loss="lat_mse")
load_climatebench_module = partial(load_model_module, task="forecasting", train_loss="mse")
load_downscaling_module = partial(
    load_model_module, task="downscaling", train_loss="mse",
    val_loss=["rmse", "pearson", "mean_bias", "mse"], test_loss=["rmse", "pearson", "mean_bias"],
    train_target_transform=None, val_target_transform=["denormalize", "denormalize", "denormalize", None],
    test_target_transform=["denormalize", "denormalize", "denormalize"])
