"""Target transforms.  Only the registry + 'denormalize' placeholder the loader API needs (validation is dead
code in the reference driver: `if False:` at examples/intermediate_downscaling.py:801)."""
TRANSFORMS_REGISTRY = {}


def register(name):
    def deco(cls):
        TRANSFORMS_REGISTRY[name] = cls
        return cls
    return deco


@register("denormalize")
class Denormalize:
    def __init__(self, data_module):
        self.norm = data_module.get_out_transforms()

    def __call__(self, x):
        import torch
        mean = torch.tensor([float(self.norm[k].mean) for k in self.norm], device=x.device).view(1, -1, 1, 1)
        std = torch.tensor([float(self.norm[k].std) for k in self.norm], device=x.device).view(1, -1, 1, 1)
        return x * std + mean
