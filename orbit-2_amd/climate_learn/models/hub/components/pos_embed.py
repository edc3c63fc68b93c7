"""Positional-embedding helpers of Res_Slim_ViT (reference: components/pos_embed.py).

These are parameter-side utilities (run at init / checkpoint load / once per step on the [L0, D] table),
not per-sample compute."""
import numpy as np
import torch
import torch.nn.functional as F


def _sincos_1d(dim, pos):
    half = dim // 2
    freq = 1.0 / np.power(10000.0, np.arange(half, dtype=np.float64) / half)
    ang = pos.reshape(-1, 1).astype(np.float64) * freq[None, :]
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size_h, grid_size_w, cls_token=False):
    """[gh*gw, D] table; first half of the channels encodes the column index, second half the row index
    (reference pos_embed.py:20-47: meshgrid(w, h) puts w first)."""
    assert embed_dim % 4 == 0, "embed_dim must be divisible by 4"
    col = np.broadcast_to(np.arange(grid_size_w, dtype=np.float64)[None, :], (grid_size_h, grid_size_w))
    row = np.broadcast_to(np.arange(grid_size_h, dtype=np.float64)[:, None], (grid_size_h, grid_size_w))
    emb = np.concatenate([_sincos_1d(embed_dim // 2, col), _sincos_1d(embed_dim // 2, row)], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros((1, embed_dim)), emb], axis=0)
    return emb


def _orig_grid(num_patches):
    oh = int((num_patches // 2) ** 0.5)   # the reference assumes W:H = 2:1 (pos_embed.py:108-111)
    return oh, 2 * oh


def interpolate_pos_embed_on_the_fly(pos_embed, patch_size, new_size=(64, 128)):
    """Bicubic (align_corners=False) resample of the [1, L0, D] table when the token-grid height differs
    (reference pos_embed.py:103-138).  A table on the device goes through the HIP kernel pair
    (`_ops.PosResFn`, csrc/image.hip orbit2_posembed_fwd / _bwd: the model's per-step path); a HOST table
    (checkpoint surgery before the model exists on a device) is resampled with torch on the CPU."""
    d = pos_embed.shape[-1]
    oh, ow = _orig_grid(pos_embed.shape[-2])
    nh, nw = new_size[0] // patch_size, new_size[1] // patch_size
    if oh == nh:
        return pos_embed
    if pos_embed.is_cuda:
        from .... import _ops  # noqa: E402  (late: the components package is imported while climate_learn is being built)
        zero = torch.zeros(d, dtype=torch.float32, device=pos_embed.device)
        out = _ops.PosResFn.apply(pos_embed.float(), zero.view(d, 1), zero, 0.0, oh, ow, nh, nw)
        return out.view(1, nh * nw, d).to(pos_embed.dtype)
    grid = pos_embed.reshape(-1, oh, ow, d).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(nh, nw), mode="bicubic", align_corners=False)
    return grid.permute(0, 2, 3, 1).flatten(1, 2)


def interpolate_pos_embed(model, checkpoint_model, new_size=(64, 128)):
    """In-place resample of checkpoint_model['pos_embed'] to the model's grid (reference pos_embed.py:75-100)."""
    if "pos_embed" not in checkpoint_model:
        return
    pe = checkpoint_model["pos_embed"]
    oh, ow = _orig_grid(pe.shape[-2])
    nh, nw = new_size[0] // model.patch_size, new_size[1] // model.patch_size
    if oh == nh:
        return
    grid = pe.reshape(-1, oh, ow, pe.shape[-1]).permute(0, 3, 1, 2)
    grid = F.interpolate(grid.float(), size=(nh, nw), mode="bicubic", align_corners=False)
    checkpoint_model["pos_embed"] = grid.permute(0, 2, 3, 1).flatten(1, 2)
