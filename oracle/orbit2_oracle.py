"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A plain-PyTorch fp32 CPU restatement of the one hot path this repo accelerates: the
Res_Slim_ViT forward (+ autograd backward), the training-step glue, the losses, AdamW and the
LR schedule of ORBIT-2's intermediate_downscaling pipeline.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this file, and only as the checker / the timed CPU
baseline -- never as something the HIP product path falls back to.

Pinned by: tests/golden/*.npz, produced by tests/golden/make_golden.py from the reference's own
modules imported in the build container (tests/test_oracle_golden.py asserts agreement <=2e-5).
Parts that stay "parity unpinned" (third-party code absent from /root/reference):
train-mode dropout / DropPath RNG streams (timm 0.9.2, ATen bernoulli), and the LPIPS network of the
`perceptual` loss (lpips package + weights absent; restated from the published algorithm, see lpips_vgg).

Written functionally over a flat {name: tensor} state dict (the reference's checkpoint key
names) so that it shares no structure with the product's nn.Module code.  Every function cites
the reference file:line (relative to /root/reference) whose arithmetic it restates.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

CONSTANT_VARS = ("land_sea_mask", "orography", "lattitude", "landcover")


# --------------------------------------------------------------------------------------
# positional embedding          src/climate_learn/models/hub/components/pos_embed.py:20-67
# --------------------------------------------------------------------------------------
def sincos_1d(dim: int, pos: np.ndarray) -> np.ndarray:
    """pos_embed.py:50-67: omega_i = 10000^(-i/(dim/2)); [sin(pos*omega) | cos(pos*omega)]."""
    half = dim // 2
    omega = 1.0 / 10000 ** (np.arange(half, dtype=np.float64) / half)
    ang = np.outer(pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def sincos_2d(dim: int, gh: int, gw: int) -> np.ndarray:
    """pos_embed.py:20-47.  NOTE the reference's naming: the first half of the channels encodes
    the *column* index (meshgrid(w, h) puts w first) and the second half the row index."""
    cols = np.tile(np.arange(gw, dtype=np.float64)[None, :], (gh, 1))
    rows = np.tile(np.arange(gh, dtype=np.float64)[:, None], (1, gw))
    return np.concatenate([sincos_1d(dim // 2, cols), sincos_1d(dim // 2, rows)], axis=1)


def pos_embed_for_grid(pos_embed: torch.Tensor, patch: int, img_size) -> torch.Tensor:
    """pos_embed.py:103-138: assumes a 2:1 (W:H) token grid; bicubic, align_corners=False, only when
    the token-grid height differs from the stored one."""
    n, d = pos_embed.shape[-2], pos_embed.shape[-1]
    oh = int((n // 2) ** 0.5)
    ow = 2 * oh
    nh, nw = img_size[0] // patch, img_size[1] // patch
    if oh == nh:
        return pos_embed
    grid = pos_embed.reshape(-1, oh, ow, d).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(nh, nw), mode="bicubic", align_corners=False)
    return grid.permute(0, 2, 3, 1).flatten(1, 2)


# --------------------------------------------------------------------------------------
# model pieces
# --------------------------------------------------------------------------------------
def patch_embed(x1: torch.Tensor, w: torch.Tensor, b: torch.Tensor, patch: int) -> torch.Tensor:
    """components/patch_embed.py:44-52: Conv2d(1, D, k=p, stride=p) -> flatten -> [B, L, D]."""
    y = F.conv2d(x1, w, b, stride=patch)
    return y.flatten(2).transpose(1, 2)


def mha_core(q, k, v, scale, pmask=None):
    """components/attention.py:72-78 (FusedAttn.NONE branch); q,k,v: [B, H, N, d].  pmask: optional multiplier on the
    attention probabilities = keep mask x 1/(1-p) of `attn_drop` (attention.py:76), supplied by tests that replicate
    the kernels' dropout masks."""
    a = (q * scale) @ k.transpose(-2, -1)
    a = a.softmax(dim=-1)
    if pmask is not None:
        a = a * pmask
    return a @ v


def variable_aggregation(x_bvld, var_query, wq, wkv, wp, bp, heads: int):
    """res_slimvit.py:205-230 + components/attention.py:132-183 (no qkv bias, attn/proj drop 0)."""
    b, v, l, d = x_bvld.shape
    x = x_bvld.permute(0, 2, 1, 3).reshape(b * l, v, d)
    hd = d // heads
    q = (var_query.expand(b * l, -1, -1) @ wq.t()).reshape(b * l, 1, heads, hd).permute(0, 2, 1, 3)
    kv = (x @ wkv.t()).reshape(b * l, v, 2, heads, hd).permute(2, 0, 3, 1, 4)
    o = mha_core(q, kv[0], kv[1], hd ** -0.5).transpose(1, 2).reshape(b * l, 1, d)
    o = o @ wp.t() + bp
    return o.reshape(b, l, d)


def attention(x, wqkv, bqkv, wp, bp, heads: int, pmask=None, omask=None):
    """components/attention.py:43-87 (tensor_par_size == 1).  Train-mode dropouts enter as explicit multipliers:
    pmask on the probabilities (attn_drop), omask on the projected output (proj_drop, :82)."""
    b, n, c = x.shape
    hd = c // heads
    qkv = (x @ wqkv.t() + bqkv).reshape(b, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    o = mha_core(qkv[0], qkv[1], qkv[2], hd ** -0.5, pmask).transpose(1, 2).reshape(b, n, c)
    o = o @ wp.t() + bp
    return o if omask is None else o * omask


def mlp(x, w1, b1, w2, b2, m1=None, m2=None):
    """components/mlp.py:57-73: fc1 -> GELU (erf) -> drop1 -> fc2 -> drop2 (masks m1 / m2 as multipliers, None = off)."""
    h = F.gelu(x @ w1.t() + b1)
    if m1 is not None:
        h = h * m1
    o = h @ w2.t() + b2
    return o if m2 is None else o * m2


def block(x, sd: Dict[str, torch.Tensor], pre: str, heads: int, masks: Optional[Dict[str, torch.Tensor]] = None):
    """components/vit_blocks.py:76-81 with ls* = Identity (init_values=None).  masks (train mode, optional): multipliers
    `attn` [B,H,N,N], `proj` [B,N,D], `fc1` [B,N,4D], `fc2` [B,N,D] and the per-sample DropPath scales `dp1`, `dp2` [B]."""
    mk = masks or {}
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5)
    a = attention(h, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"],
                  sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"], heads, mk.get("attn"), mk.get("proj"))
    if "dp1" in mk:
        a = a * mk["dp1"].view(-1, 1, 1)
    x = x + a
    h = F.layer_norm(x, (d,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5)
    m = mlp(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"],
            sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"], mk.get("fc1"), mk.get("fc2"))
    if "dp2" in mk:
        m = m * mk["dp2"].view(-1, 1, 1)
    return x + m


def unpatchify(x, img_size, patch: int, scaling: int, out_channels: int):
    """res_slimvit.py:167-179.  NOTE: the per-token block edge is `patch` (not patch*scaling): the
    C*(s*p)^2 features of a token are laid out as (p, p, C) on a token grid that is `scaling`
    times finer -- i.e. the token sequence itself is re-read as an (h*s/p) x (w*s/p) grid."""
    p, c = patch, out_channels
    h = img_size[0] * scaling // p
    w = img_size[1] * scaling // p
    x = x.reshape(x.shape[0], h, w, p, p, c)
    x = torch.einsum("nhwpqc->nchpwq", x)
    return x.reshape(x.shape[0], c, h * p, w * p)


def path2(x_sel, w0, b0, w3, b3, mag: int):
    """res_slimvit.py:107-112: conv3x3 -> GELU -> PixelShuffle(mag) -> conv3x3."""
    y = F.gelu(F.conv2d(x_sel, w0, b0, padding=1))
    y = F.pixel_shuffle(y, mag)
    return F.conv2d(y, w3, b3, padding=1)


class Config:
    """Hyper-parameters of one Res_Slim_ViT instance (res_slimvit.py:22-43) + run-time data_config."""

    def __init__(self, default_vars: Sequence[str], img_size, out_channels: int, embed_dim: int, depth: int,
                 decoder_depth: int, num_heads: int, patch_size: int = 2, superres_mag: int = 4,
                 cnn_ratio: int = 4, mlp_ratio: float = 4.0, spatial_resolution: float = 0.0):
        self.default_vars = list(default_vars)
        self.img_size = tuple(img_size)
        self.out_channels = out_channels
        self.embed_dim = embed_dim
        self.depth = depth
        self.decoder_depth = decoder_depth
        self.num_heads = num_heads
        self.patch_size = patch_size
        self.superres_mag = superres_mag
        self.cnn_ratio = cnn_ratio
        self.mlp_ratio = mlp_ratio
        self.spatial_resolution = spatial_resolution


def init_state_dict(cfg: Config, n_in: int, seed: int = 0, init_grid=None, fast: bool = False) -> Dict[str, torch.Tensor]:
    """Random-init weights with the reference's shapes and init laws (res_slimvit.py:125-145:
    Linear trunc_normal(0.02)/bias 0, LayerNorm 1/0, conv default, var_embed/var_query zeros,
    pos_embed sincos).  RNG stream differs from the reference's (parity unpinned for init values)."""
    g = torch.Generator().manual_seed(seed)
    d, c, p, s = cfg.embed_dim, cfg.out_channels, cfg.patch_size, cfg.superres_mag
    grid = init_grid or cfg.img_size
    gh, gw = grid[0] // p, grid[1] // p
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, o, i, bias=True):
        if fast:        # timing-only init (bench cpu_baseline): plain N(0, 0.02), no truncation
            w = torch.randn(o, i, generator=g) * 0.02
        else:
            w = torch.empty(o, i)
            torch.nn.init.trunc_normal_(w, std=0.02, generator=g)
        sd[name + ".weight"] = w
        if bias:
            sd[name + ".bias"] = torch.zeros(o)

    def conv(name, o, i, k):
        fan_in = i * k * k
        bound = 1.0 / math.sqrt(fan_in)
        sd[name + ".weight"] = (torch.rand(o, i, k, k, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * bound

    nv = len(cfg.default_vars)
    sd["var_embed"] = torch.zeros(1, nv, d)
    sd["var_query"] = torch.zeros(1, 1, d)
    sd["pos_embed"] = torch.from_numpy(sincos_2d(d, gh, gw)).float().unsqueeze(0)
    lin("spatial_embed", d, 1)
    for i in range(nv):
        conv("token_embeds.%d.proj" % i, d, 1, p)
    lin("var_agg.q", d, d, bias=False)
    lin("var_agg.kv", 2 * d, d, bias=False)
    lin("var_agg.proj", d, d)
    hid = int(d * cfg.mlp_ratio)
    for i in range(cfg.depth):
        pre = "blocks.%d." % i
        sd[pre + "norm1.weight"] = torch.ones(d)
        sd[pre + "norm1.bias"] = torch.zeros(d)
        lin(pre + "attn.qkv", 3 * d, d)
        lin(pre + "attn.proj", d, d)
        sd[pre + "norm2.weight"] = torch.ones(d)
        sd[pre + "norm2.bias"] = torch.zeros(d)
        lin(pre + "mlp.fc1", hid, d)
        lin(pre + "mlp.fc2", d, hid)
    sd["norm.weight"] = torch.ones(d)
    sd["norm.bias"] = torch.zeros(d)
    conv("path2.0", cfg.cnn_ratio * s * s, c + 4, 3)
    conv("path2.3", c, cfg.cnn_ratio, 3)
    for i in range(cfg.decoder_depth):
        lin("head.%d" % (2 * i), d, d)
    lin("head.%d" % (2 * cfg.decoder_depth), c * (s * p) ** 2, d)
    conv("conv_out", c, c, 3)
    return sd


def forward(sd: Dict[str, torch.Tensor], cfg: Config, x: torch.Tensor, in_variables: List[str],
            out_variables: List[str], masks: Optional[Dict] = None) -> torch.Tensor:
    """res_slimvit.py:312-338 (forward) with :245-299 (forward_encoder).  Eval mode unless `masks` supplies the
    train-mode dropout / DropPath multipliers ({"pos": [B,L,D], "blocks.i": {...}}, see block())."""
    if x.dim() == 5:
        x = x.flatten(1, 2)
    p, d = cfg.patch_size, cfg.embed_dim
    # residual branch inputs: out vars then the four constants   (res_slimvit.py:302-310)
    idx = [in_variables.index(v) for v in out_variables] + [in_variables.index(v) for v in CONSTANT_VARS]
    res = path2(x[:, idx], sd["path2.0.weight"], sd["path2.0.bias"], sd["path2.3.weight"], sd["path2.3.bias"],
                cfg.superres_mag)
    # per-variable tokenisation + variable embedding      (res_slimvit.py:250-262)
    ids = [cfg.default_vars.index(v) for v in in_variables]
    toks = [patch_embed(x[:, i:i + 1], sd["token_embeds.%d.proj.weight" % vid],
                        sd["token_embeds.%d.proj.bias" % vid], p) for i, vid in enumerate(ids)]
    t = torch.stack(toks, dim=1) + sd["var_embed"][:, ids].unsqueeze(2)
    t = variable_aggregation(t, sd["var_query"], sd["var_agg.q.weight"], sd["var_agg.kv.weight"],
                             sd["var_agg.proj.weight"], sd["var_agg.proj.bias"], cfg.num_heads)
    # positional + resolution embedding                   (res_slimvit.py:270-284)
    t = t + pos_embed_for_grid(sd["pos_embed"], p, cfg.img_size)
    res_km = torch.tensor([float(cfg.spatial_resolution)], dtype=t.dtype)
    t = t + (sd["spatial_embed.weight"] @ res_km + sd["spatial_embed.bias"]).view(1, 1, d)
    if masks is not None and "pos" in masks:          # pos_drop (res_slimvit.py:284)
        t = t * masks["pos"]
    for i in range(cfg.depth):
        t = block(t, sd, "blocks.%d." % i, cfg.num_heads, None if masks is None else masks.get("blocks.%d" % i))
    t = F.layer_norm(t, (d,), sd["norm.weight"], sd["norm.bias"], 1e-5)
    # decoder head                                          (res_slimvit.py:115-120,326)
    for i in range(cfg.decoder_depth):
        t = F.gelu(t @ sd["head.%d.weight" % (2 * i)].t() + sd["head.%d.bias" % (2 * i)])
    k = 2 * cfg.decoder_depth
    t = t @ sd["head.%d.weight" % k].t() + sd["head.%d.bias" % k]
    img = unpatchify(t, cfg.img_size, p, cfg.superres_mag, cfg.out_channels)
    img = F.conv2d(img, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    return img + res[:, :, : img.shape[2], : img.shape[3]]


# --------------------------------------------------------------------------------------
# training-step glue            examples/intermediate_downscaling.py:267-306
# --------------------------------------------------------------------------------------
def clip_replace_constant(y, yhat, out_variables):
    """:267-278: clamp the precipitation channel at 0; constant channels copy the ground truth."""
    pi = out_variables.index("total_precipitation_24hr")
    chans = []
    for i, name in enumerate(out_variables):
        if name in CONSTANT_VARS:
            chans.append(y[:, i, : yhat.shape[2], : yhat.shape[3]])
        elif i == pi:
            chans.append(yhat[:, i].clamp(min=0.0))
        else:
            chans.append(yhat[:, i])
    return torch.stack(chans, dim=1)


def crop_target(y, yhat):
    """:295-296: target cropped to the prediction's top-left H x W."""
    return y[:, :, : yhat.shape[2], : yhat.shape[3]]


# --------------------------------------------------------------------------------------
# losses                        src/climate_learn/metrics/functional.py
# --------------------------------------------------------------------------------------
def lat_weights(lat: np.ndarray, rows: Optional[int] = None) -> torch.Tensor:
    """metrics/metrics.py:58-65: cos(lat)/mean(cos(lat)) shaped (1,1,H,1); `rows` crops to the
    prediction height (intended semantics)."""
    w = np.cos(np.deg2rad(np.asarray(lat, dtype=np.float64)))
    w = w / w.mean()
    if rows is not None:
        w = w[:rows]
    return torch.from_numpy(w).view(1, 1, -1, 1).float()


def _channel_weights(pred, var_names, var_weights):
    if var_names is None:
        return None
    assert len(var_names) == pred.shape[1], "Number of variable names must match channel dimension"
    w = torch.tensor([float(var_weights.get(v, 1.0)) for v in var_names], dtype=pred.dtype)
    return w.view(1, -1, 1, 1)


def _reduce(err, aggregate_only):
    if aggregate_only:
        return err.mean()
    return torch.cat((err.mean([0, 2, 3]), err.mean().unsqueeze(0)))


def mse(pred, target, var_names=None, var_weights=None, aggregate_only=False, lat_w=None):
    """functional.py:173-202."""
    err = (pred - target).square()
    if lat_w is not None:
        err = err * lat_w
    cw = _channel_weights(pred, var_names, var_weights)
    if cw is not None:
        err = err * cw
    return _reduce(err, aggregate_only)


def bayesian_tv(pred, target, var_names=None, var_weights=None, aggregate_only=False, lat_w=None):
    """functional.py:117-167: squared error + 0.02*(|dv| + |dh| + 0.7|d1| + 0.7|d2|) of pred; each
    difference map is zero-padded back to H x W on the side the reference pads
    (dv,dh,d1: bottom/right; d2 = pred[i+1,j]-pred[i,j+1] lands at column j+1)."""
    err = (pred - target).square()
    dv = F.pad((pred[:, :, 1:, :] - pred[:, :, :-1, :]).abs(), (0, 0, 0, 1))
    dh = F.pad((pred[:, :, :, 1:] - pred[:, :, :, :-1]).abs(), (0, 1))
    d1 = F.pad((pred[:, :, 1:, 1:] - pred[:, :, :-1, :-1]).abs(), (0, 1, 0, 1))
    d2 = F.pad((pred[:, :, 1:, :-1] - pred[:, :, :-1, 1:]).abs(), (1, 0, 0, 1))
    err = err + 0.02 * (dv + dh + 0.7 * d1 + 0.7 * d2)
    if lat_w is not None:
        err = err * lat_w
    cw = _channel_weights(pred, var_names, var_weights)
    if cw is not None:
        err = err * cw
    return _reduce(err, aggregate_only)


def image_gradient(pred, target, var_names=None, var_weights=None):
    """functional.py:59-114 with torchmetrics.functional.image.image_gradients (un-vendored; restated
    from its documented definition: forward differences, last row / column zero).  Parity unpinned."""
    def grads(img):
        dy = F.pad(img[:, :, 1:, :] - img[:, :, :-1, :], (0, 0, 0, 1))
        dx = F.pad(img[:, :, :, 1:] - img[:, :, :, :-1], (0, 1))
        return dy, dx
    e1 = (pred - target).square()
    dy, dx = grads(target)
    hy, hx = grads(pred)
    e2 = ((dx - hx).abs() + (dy - hy).abs()).mean()
    cw = _channel_weights(pred, var_names, var_weights)
    if cw is not None:
        e1 = e1 * cw
        e2 = e2 * cw
    return e1.mean() + 0.1 * e2.mean()


# --------------------------------------------------------------------------------------
# perceptual loss = L1 + 0.5 * mean_b LPIPS-VGG16          (metrics/functional.py:17-33, metrics.py:119-187)
# LPIPS lives in the third-party `lpips` package (unpinned in the reference's pyproject, weights not in the
# tree) -> PARITY UNPINNED.  Restated from the published algorithm (Zhang et al. 2018, lpips 0.1.x, net='vgg',
# version 0.1, spatial=False, eval mode => the lin layers' Dropout is inactive):
#   ScalingLayer (x - shift) / scale ; torchvision VGG16 `features` taps after relu1_2, relu2_2, relu3_3,
#   relu4_3, relu5_3 ; per tap: unit-normalise over channels (x / (||x||_2 + 1e-10)), squared difference,
#   1x1 conv with the learned non-negative `lin` weights (no bias), spatial mean ; sum over the 5 taps.
# --------------------------------------------------------------------------------------
VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)
VGG16_TAPS = (1, 3, 6, 9, 12)            # conv indices (0-based) whose ReLU output is an LPIPS tap
LPIPS_SHIFT = (-0.030, -0.088, -0.188)
LPIPS_SCALE = (0.458, 0.448, 0.450)


def init_lpips_weights(seed: int = 0) -> Dict[str, torch.Tensor]:
    """random stand-in weights (He-normal convs, small positive lins): `conv{i}.weight [Co,Ci,3,3]`,
    `conv{i}.bias`, `lin{k}.weight [C]` -- the synthetic-throughput configuration of SURVEY 8(d).5"""
    g = torch.Generator().manual_seed(seed)
    sd, cin, i = {}, 3, 0
    for c in VGG16_CFG:
        if c == "M":
            continue
        sd["conv%d.weight" % i] = torch.randn(c, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
        sd["conv%d.bias" % i] = torch.randn(c, generator=g) * 0.05
        cin = c
        i += 1
    for k, ci in enumerate(VGG16_TAPS):
        ch = sd["conv%d.weight" % ci].shape[0]
        sd["lin%d.weight" % k] = torch.rand(ch, generator=g) * (2.0 / ch)
    return sd


def lpips_vgg(x0: torch.Tensor, x1: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """[B,3,H,W] x2 -> [B] distances"""
    shift = torch.tensor(LPIPS_SHIFT, dtype=x0.dtype).view(1, 3, 1, 1)
    scale = torch.tensor(LPIPS_SCALE, dtype=x0.dtype).view(1, 3, 1, 1)

    def feats(x):
        h, out, i = (x - shift) / scale, [], 0
        for c in VGG16_CFG:
            if c == "M":
                h = F.max_pool2d(h, 2, 2)
                continue
            h = F.relu(F.conv2d(h, sd["conv%d.weight" % i], sd["conv%d.bias" % i], padding=1))
            if i in VGG16_TAPS:
                out.append(h)
            i += 1
        return out

    val = 0.0
    for k, (f0, f1) in enumerate(zip(feats(x0), feats(x1))):
        n0 = f0 / (f0.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        n1 = f1 / (f1.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        d = ((n0 - n1) ** 2 * sd["lin%d.weight" % k].view(1, -1, 1, 1)).sum(1)
        val = val + d.mean((1, 2))
    return val


def perceptual(pred, target, lpips_sd):
    """F.l1_loss(pred, target) + 0.5 * mean(LPIPS(pred, target))     (metrics/functional.py:30)"""
    return (pred - target).abs().mean() + 0.5 * lpips_vgg(pred, target, lpips_sd).mean()


# --------------------------------------------------------------------------------------
# evaluation metrics (metrics/functional.py:236-324) -- pinned by tests/golden/eval_metrics.npz
# --------------------------------------------------------------------------------------
def rmse(pred, target, aggregate_only=False, lat_w=None):
    """:236-255: sqrt of the per-(b,c) spatial mean of the (latitude-weighted) squared error, mean over b then c"""
    err = (pred - target).square()
    if lat_w is not None:
        err = err * lat_w
    return _reduce_vec(err.mean([2, 3]).sqrt().mean(0), aggregate_only)


def pearson(pred, target, aggregate_only=False):
    """:294-308: cosine similarity of the mean-removed [C, B*H*W] fields"""
    C = pred.shape[1]
    p = pred.transpose(0, 1).reshape(C, -1)
    t = target.transpose(0, 1).reshape(C, -1)
    p = p - p.mean(1, keepdim=True)
    t = t - t.mean(1, keepdim=True)
    return _reduce_vec(F.cosine_similarity(p, t), aggregate_only)


def mean_bias(pred, target, aggregate_only=False):
    """:311-324: mean(target) - mean(pred) per channel"""
    return _reduce_vec(target.mean((0, 2, 3)) - pred.mean((0, 2, 3)), aggregate_only)


def mae(pred, target, aggregate_only=False, lat_w=None):
    """:219-232: mean absolute (latitude-weighted) error per channel and overall"""
    err = (pred - target).abs()
    if lat_w is not None:
        err = err * lat_w
    per = err.mean([0, 2, 3])
    return err.mean() if aggregate_only else torch.cat((per, err.mean().unsqueeze(0)))


def acc(pred, target, climatology, aggregate_only=False, lat_w=None):
    """:259-291: anomaly correlation; channels centred by their unweighted mean, latitude-weighted sums (the
    reference's mask branch is overwritten by the unmasked sums, so there is no mask here)"""
    w = lat_w if lat_w is not None else torch.ones(1, 1, pred.shape[2], 1)
    a, b = pred - climatology, target - climatology
    per = []
    for i in range(pred.shape[1]):
        pa, pb = a[:, i] - a[:, i].mean(), b[:, i] - b[:, i].mean()
        ww = w[0]                                                     # [1,H,1] against [B,H,W]
        per.append((ww * pa * pb).sum() / ((ww * pa.square()).sum() * (ww * pb.square()).sum()).sqrt())
    return _reduce_vec(torch.stack(per), aggregate_only)


def _reduce_vec(per_channel, aggregate_only):
    agg = per_channel.mean()
    return agg if aggregate_only else torch.cat((per_channel, agg.unsqueeze(0)))


LOSSES = {"mse": mse, "bayesian_tv": bayesian_tv}


def training_loss(sd, cfg, x, y, in_variables, out_variables, loss_name="bayesian_tv", var_weights=None,
                  lat=None, lpips_sd=None, masks=None):
    """training_step (:281-306): forward, clip, crop, loss (aggregate)."""
    pred = forward(sd, cfg, x, in_variables, out_variables, masks)
    yhat = clip_replace_constant(y, pred, out_variables)
    tgt = crop_target(y, yhat)
    if loss_name == "perceptual":
        return perceptual(yhat, tgt, lpips_sd)
    if loss_name == "lat_mse":
        return mse(yhat, tgt, out_variables, var_weights or {}, True, lat_weights(lat, yhat.shape[2]))
    if loss_name == "perceptual_lat_mse":       # BASELINE configs[4]: perceptual + intended lat_mse
        return perceptual(yhat, tgt, lpips_sd) + mse(yhat, tgt, out_variables, var_weights or {}, True,
                                                     lat_weights(lat, yhat.shape[2]))
    return LOSSES[loss_name](yhat, tgt, out_variables, var_weights or {}, True)


# --------------------------------------------------------------------------------------
# optimizer + schedule
# --------------------------------------------------------------------------------------
def adamw_step(p, g, m, v, step: int, lr: float, beta1: float, beta2: float, eps: float, wd: float):
    """torch.optim.AdamW single-tensor math (utils/loaders.py:398-399 selects it): decoupled decay,
    bias-corrected moments; in place on p, m, v.  `step` is 1-based."""
    p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def warmup_cosine_lr(epoch: int, base_lr: float, warmup_epochs: int, max_epochs: int,
                     warmup_start_lr: float, eta_min: float) -> float:
    """models/lr_scheduler.py:93-115 (closed form; the chainable form :39-91 gives the same values)."""
    if epoch < warmup_epochs:
        return warmup_start_lr + epoch * (base_lr - warmup_start_lr) / max(1, warmup_epochs - 1)
    return eta_min + 0.5 * (base_lr - eta_min) * (
        1 + math.cos(math.pi * (epoch - warmup_epochs) / (max_epochs - warmup_epochs)))


# --------------------------------------------------------------------------------------
# work model                    SURVEY.md 8(d)
# --------------------------------------------------------------------------------------
def forward_flops(L, V, D, depth, dd, C, h, w, heads, p=2, s=4, cr=4, r=4, folded_varagg=False):
    """Dense-formulation forward FLOPs per sample.  folded_varagg=True replaces the
    kv GEMM by what the folded kernel executes (5 MACs per (token, var, channel) + scores)."""
    pe = 2 * L * V * p * p * D
    if folded_varagg:
        va = 2 * L * V * D * 5 + 2 * L * V * heads * 5 + 2 * L * D * D      # fold + proj
    else:
        va = 2 * L * V * D * 2 * D + 2 * (2 * L * D * D) + 4 * L * V * D
        va += pe
    blk = depth * (2 * L * D * 3 * D + 4 * L * L * D + 2 * L * D * D + 4 * L * D * r * D)
    head = dd * 2 * L * D * D + 2 * L * D * C * (p * s) ** 2
    convs = 2 * h * w * (C + 4) * (cr * s * s) * 9 + 2 * (16 * h * w) * cr * C * 9 + 2 * (16 * h * w) * C * C * 9
    return va + blk + head + convs
