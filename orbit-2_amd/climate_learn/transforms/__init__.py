"""Target transforms: the registry + 'denormalize' (reference transforms/denormalize.py:15-33: x*std + mean per output
channel, precipitation variables left as they are).  Used by the validation / test metrics and the tiled-inference
path (validation itself is dead code in the reference training driver: `if False:` at :801)."""
TRANSFORMS_REGISTRY = {}


def register(name):
    def deco(cls):
        TRANSFORMS_REGISTRY[name] = cls
        return cls
    return deco


@register("denormalize")
class Denormalize:
    def __init__(self, data_module):
        from ..data.processing.era5_constants import PRECIP_VARIABLES
        self.norm = data_module.get_out_transforms()
        if self.norm is None:
            raise RuntimeError("norm was 'None', did you setup the data module?")
        self.mean = [0.0 if k in PRECIP_VARIABLES else float(self.norm[k].mean) for k in self.norm]
        self.std = [1.0 if k in PRECIP_VARIABLES else float(self.norm[k].std) for k in self.norm]

    def __call__(self, x):
        """x: [B,C,H,W] or [C,H,W] (torchvision Normalize semantics: channel = dim -3)"""
        import torch
        shape = (-1, 1, 1)
        mean = torch.tensor(self.mean, device=x.device, dtype=x.dtype).view(shape)
        std = torch.tensor(self.std, device=x.device, dtype=x.dtype).view(shape)
        return x * std + mean
