"""Self-attention and the variable-aggregation cross-attention of Res_Slim_ViT
(reference: components/attention.py:12-87 and :98-183), on the HIP kernels."""
import torch
import torch.nn as nn

from .... import _ops
from ....utils.fused_attn import FusedAttn
from ....dist import tp as _tp
from .mlp import HipLinear


class Attention(nn.Module):
    def __init__(self, dim, fused_attn=FusedAttn.HIP, num_heads=8, qkv_bias=False, qk_norm=False, proj_bias=True,
                 attn_drop=0.0, proj_drop=0.0, norm_layer=nn.LayerNorm, tensor_par_size=1, tensor_par_group=None):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        assert num_heads % tensor_par_size == 0, "model heads % tensor parallel size must be 0"
        assert not qk_norm, "qk_norm is not used by Res_Slim_ViT"
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.fused_attn = fused_attn
        self.tensor_par_size, self.tensor_par_group = tensor_par_size, tensor_par_group
        # head-split: this rank holds num_heads / tensor_par_size heads (reference attention.py:36-40)
        self.qkv = HipLinear(dim, dim * 3 // tensor_par_size, bias=qkv_bias)
        self.proj = HipLinear(dim // tensor_par_size, dim, bias=proj_bias)
        self.attn_drop_p, self.proj_drop_p = float(attn_drop), float(proj_drop)

    def attn_p(self):
        """P-dropout probability actually applied (CK semantics: also in eval mode)."""
        if self.training or FusedAttn(self.fused_attn).dropout_in_eval:
            return self.attn_drop_p
        return 0.0

    def forward(self, x):
        tp, grp = self.tensor_par_size, self.tensor_par_group
        if tp > 1:
            x = _tp.IdentityFwdAllReduceBwd.apply(x, grp)
        qkv = self.qkv(x)
        o = _ops.AttnCoreFn.apply(qkv, self.num_heads // tp, self.attn_p(), _tp.group_rank(grp) if tp > 1 else 0)
        y = self.proj(o, self.proj_drop_p if self.training else 0.0)
        return _tp.AllReduceFwdIdentityBwd.apply(y, grp) if tp > 1 else y


class VariableMapping_Attention(nn.Module):
    """Owns q / kv / proj under the reference's names.  Its arithmetic is executed by Res_Slim_ViT through the
    folded tables (see csrc/varagg.hip): q and kv never run as per-token GEMMs."""

    def __init__(self, dim, fused_attn=FusedAttn.HIP, num_heads=8, qkv_bias=False, qk_norm=False, proj_bias=True,
                 attn_drop=0.0, proj_drop=0.0, norm_layer=nn.LayerNorm, tensor_par_size=1, tensor_par_group=None):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        assert num_heads % tensor_par_size == 0, "model heads % tensor parallel size must be 0"
        assert not qkv_bias and not qk_norm and attn_drop == 0.0 and proj_drop == 0.0, \
            "the folded kernel implements the configuration Res_Slim_ViT instantiates (res_slimvit.py:78)"
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.tensor_par_size, self.tensor_par_group = tensor_par_size, tensor_par_group
        self.q = HipLinear(dim, dim // tensor_par_size, bias=False)
        self.kv = HipLinear(dim, dim * 2 // tensor_par_size, bias=False)
        for p in (self.q.weight, self.kv.weight):   # used in fp32 table algebra, not in bf16 GEMMs
            p._o2_lowp = False
        self.proj = HipLinear(dim // tensor_par_size, dim, bias=proj_bias)
