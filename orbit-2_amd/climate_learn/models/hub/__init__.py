from .utils import MODEL_REGISTRY
from .res_slimvit import Res_Slim_ViT
