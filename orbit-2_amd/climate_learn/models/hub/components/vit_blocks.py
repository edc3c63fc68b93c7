"""Transformer block of Res_Slim_ViT (reference: components/vit_blocks.py:25-81).

forward() is ONE autograd node (climate_learn._ops.BlockFn): LN1 -> qkv GEMM -> flash attention -> proj GEMM
(+bias, dropout, DropPath, residual fused) -> LN2 -> fc1 GEMM (+GELU, dropout) -> fc2 GEMM (+dropout, DropPath,
residual), with a hand-written backward of the same kernels.  `recompute=True` is the MI355X counterpart of
the reference's per-Block activation checkpointing (examples/intermediate_downscaling.py:633-637): only the
block input is kept and the forward kernels are replayed in backward."""
import torch
import torch.nn as nn

from .... import _ops
from ....utils.fused_attn import FusedAttn
from .attention import Attention
from .mlp import Mlp


class HipLayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        assert eps == 1e-5
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.weight._o2_lowp = True
        self.bias._o2_lowp = True

    def forward(self, x):
        return _ops.LayerNormFn.apply(x, self.weight, self.bias)


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5, inplace=False):
        super().__init__()
        raise NotImplementedError("LayerScale is unused by Res_Slim_ViT (init_values=None)")


class Block(nn.Module):
    def __init__(self, dim, num_heads, fused_attn=FusedAttn.HIP, mlp_ratio=4.0, qkv_bias=False, qk_norm=False,
                 proj_bias=True, proj_drop=0.0, attn_drop=0.0, init_values=None, drop_path=0.0, act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, mlp_layer=Mlp, tensor_par_size=1, tensor_par_group=None):
        super().__init__()
        assert init_values is None and act_layer is nn.GELU and mlp_layer is Mlp
        assert qkv_bias and proj_bias, "the fused block kernel path expects biased qkv/proj (res_slimvit.py:93)"
        self.norm1 = HipLayerNorm(dim)
        self.attn = Attention(dim, fused_attn=fused_attn, num_heads=num_heads, qkv_bias=qkv_bias, qk_norm=qk_norm,
                              proj_bias=proj_bias, attn_drop=attn_drop, proj_drop=proj_drop,
                              tensor_par_size=tensor_par_size, tensor_par_group=tensor_par_group)
        self.norm2 = HipLayerNorm(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), bias=proj_bias, drop=proj_drop,
                       tensor_par_size=tensor_par_size, tensor_par_group=tensor_par_group)
        self.drop_path = float(drop_path)
        self.recompute = False
        self.tensor_par_group = tensor_par_group if tensor_par_size > 1 else None

    def forward(self, x):
        tr = self.training
        cfg = {
            "heads": self.attn.num_heads,
            "attn_drop": self.attn.attn_p(),
            "proj_drop": self.attn.proj_drop_p if tr else 0.0,
            "mlp_drop": self.mlp.drop if tr else 0.0,
            "drop_path": self.drop_path if tr else 0.0,
            "recompute": self.recompute,
            "tp_group": self.tensor_par_group,
        }
        a, m = self.attn, self.mlp
        x = _ops.unit_enter(self, x)   # parameter-sharding engine: gather this Block's weights (no-op otherwise)
        y = _ops.BlockFn.apply(x, cfg, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias,
                               a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight,
                               m.fc1.bias, m.fc2.weight, m.fc2.bias)
        return _ops.unit_exit(self, y)
