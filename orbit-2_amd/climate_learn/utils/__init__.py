from .fused_attn import FusedAttn
from .checkpoint import load_checkpoint, load_pretrained_weights
from . import visualize
from .visualize import tiled_predict, tile_windows, visualize_at_index
from . import loaders as _loaders

for _n in dir(_loaders):                      # every load_* factory, as the reference's utils package re-exports them
    if _n.startswith("load_"):
        globals()[_n] = getattr(_loaders, _n)
