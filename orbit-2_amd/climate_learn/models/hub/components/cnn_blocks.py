"""Names the reference driver imports from here (examples/intermediate_downscaling.py:34-39).  The U-Net /
ResNet baselines that use them are outside the Res_Slim_ViT hot path (OUT OF SCOPE), so these
are importable placeholders that refuse construction."""
import torch.nn as nn


class _OutOfScope(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError(
            "%s belongs to the legacy CNN baselines, which this MI355X build of the Res_Slim_ViT hot path does "
            "not include" % type(self).__name__)


class PeriodicPadding2D(_OutOfScope):
    pass


class PeriodicConv2D(_OutOfScope):
    pass


class ResidualBlock(_OutOfScope):
    pass


class DownBlock(_OutOfScope):
    pass


class MiddleBlock(_OutOfScope):
    pass


class UpBlock(_OutOfScope):
    pass
