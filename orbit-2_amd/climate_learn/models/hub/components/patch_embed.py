"""Per-variable patch embedding parameters (reference: components/patch_embed.py:22-53).

In the HIP path the V per-variable Conv2d(1, D, p, p) are never run one by one: Res_Slim_ViT folds them,
together with the variable-aggregation attention, into two small tables + one per-token kernel
(csrc/varagg.hip).  This module owns the parameters under the reference's names (`proj.weight [D,1,p,p]`,
`proj.bias [D]`) and offers a stand-alone forward built on the same GEMM kernel."""
import math

import torch
import torch.nn as nn

from .... import _hip


class _ConvParams(nn.Module):
    def __init__(self, in_ch, out_ch, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_ch, in_ch, k, k))
        self.bias = nn.Parameter(torch.empty(out_ch))
        bound = 1.0 / math.sqrt(in_ch * k * k)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True,
                 bias=True):
        super().__init__()
        two = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        self.img_size, self.patch_size = two(img_size), two(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        assert norm_layer is None and bias, "only the configuration Res_Slim_ViT uses is implemented"
        self.proj = _ConvParams(in_chans, embed_dim, self.patch_size[0])

    def forward(self, x):
        """[B, C, H, W] fp32 -> [B, L, D] bf16 via patch gather + the MFMA GEMM (K padded to 64)."""
        B, C, H, W = x.shape
        p = self.patch_size[0]
        D = self.proj.weight.shape[0]
        cols = x.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(-1, C * p * p)
        K = (cols.shape[1] + 63) // 64 * 64
        a = torch.zeros(cols.shape[0], K, dtype=torch.bfloat16, device=x.device)
        a[:, : cols.shape[1]] = cols
        wmat = torch.zeros(D, K, dtype=torch.bfloat16, device=x.device)
        wmat[:, : cols.shape[1]] = self.proj.weight.detach().reshape(D, -1)
        out = torch.empty(cols.shape[0], D, dtype=torch.bfloat16, device=x.device)
        _hip.gemm(a, wmat, out, cols.shape[0], D, K, K, K, D, bias=self.proj.bias.detach().to(torch.bfloat16))
        return out.view(B, -1, D)
