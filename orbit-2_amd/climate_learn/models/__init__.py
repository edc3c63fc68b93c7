from .hub import MODEL_REGISTRY
from .lr_scheduler import LinearWarmupCosineAnnealingLR
