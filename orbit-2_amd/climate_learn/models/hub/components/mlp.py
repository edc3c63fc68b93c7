"""ViT MLP (reference: components/mlp.py:22-73): fc1 -> GELU(erf) -> Dropout -> fc2 -> Dropout, as two MFMA
GEMMs with GELU+dropout fused into the first epilogue and dropout into the second."""
import torch
import torch.nn as nn

from .... import _ops


class HipLinear(nn.Module):
    """Parameter holder with nn.Linear's names/shapes; compute goes through the HIP GEMM."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        nn.init.trunc_normal_(self.weight, std=0.02)
        self.weight._o2_lowp = True
        if self.bias is not None:
            self.bias._o2_lowp = True

    def forward(self, x, p_drop=0.0, residual=None):
        return _ops.LinearFn.apply(x, self.weight, self.bias, p_drop, residual)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, norm_layer=None,
                 bias=True, drop=0.0, use_conv=False, tensor_par_size: int = 1, tensor_par_group=None):
        super().__init__()
        assert act_layer is nn.GELU and norm_layer is None and not use_conv
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        assert hidden_features % tensor_par_size == 0
        self.tensor_par_size, self.tensor_par_group = tensor_par_size, tensor_par_group
        # hidden units are divided over the tensor-parallel ranks (reference mlp.py:50-55)
        self.fc1 = HipLinear(in_features, hidden_features // tensor_par_size, bias=bias)
        self.fc2 = HipLinear(hidden_features // tensor_par_size, out_features, bias=bias)
        self.drop = float(drop)

    def forward(self, x):
        from ....dist import tp as _tp
        p = self.drop if self.training else 0.0
        cfg = {"ln": False, "p_mid": p, "p_out": p}
        if self.tensor_par_size > 1:
            x = _tp.IdentityFwdAllReduceBwd.apply(x, self.tensor_par_group)
        y = _ops.ChainFn.apply(x, cfg, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        return _tp.AllReduceFwdIdentityBwd.apply(y, self.tensor_par_group) if self.tensor_par_size > 1 else y
