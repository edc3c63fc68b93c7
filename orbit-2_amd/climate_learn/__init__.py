"""climate_learn -- MI355X-native (gfx950) build of ORBIT-2's intermediate_downscaling hot path.

Mirrors the reference package surface the driver uses (`import climate_learn as cl`): the load_* factories,
`cl.data.IterDataModule`, the model / metrics registries, `FusedAttn`.  The compute goes through
liborbit2_hip.so (include/orbit2_hip.h) -- every fused op and every GEMM of the step, forward and backward; no
vendor-library GEMM is called.  There is no CPU fallback."""
from .utils.fused_attn import FusedAttn
from .utils import loaders as _loaders

# the factory functions the reference re-exports at package level (src/climate_learn/__init__.py:1-11)
_FACTORIES = ("load_model_module load_forecasting_module load_climatebench_module load_downscaling_module "
              "load_architecture load_optimizer load_lr_scheduler load_loss load_transform").split()
globals().update({_n: getattr(_loaders, _n) for _n in _FACTORIES})
from . import data
from .dist.dp_engine import CommStats, HipDataParallel
from .dist.fsdp_engine import HipFullyShardedDataParallel
from .optim import HipAdamW, HipGradScaler
from ._ops import manual_seed
from .graphs import GraphedTrainStep
