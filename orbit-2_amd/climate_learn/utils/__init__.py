from .fused_attn import FusedAttn
from .checkpoint import load_checkpoint, load_pretrained_weights
from . import visualize
from .visualize import tiled_predict, tile_windows, visualize_at_index
from .loaders import (
    load_model_module,
    load_forecasting_module,
    load_downscaling_module,
    load_climatebench_module,
    load_architecture,
    load_optimizer,
    load_lr_scheduler,
    load_loss,
    load_transform,
)
