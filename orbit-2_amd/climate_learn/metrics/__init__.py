from .utils import METRICS_REGISTRY, MetricsMetaInfo
from .metrics import *  # noqa: F401,F403
from .functional import mse, bayesian_tv, image_gradient
