"""Whole-model GPU tests of the BASELINE configurations that round 1 only covered kernel by kernel:
configs[1] interm_117m (D1024 / 16 heads of 64 / depth 8, 23 -> 3 variables, 32x64 -> 128x256): forward + loss + backward
against the CPU oracle; configs[3] interm_10b (D8192 / 32 heads of 256 / depth 11, 9.47 B parameters): one training step
with activation recompute at batch 1 -- finite loss, recompute == saved-activation gradients bit for bit on a Block,
batch independence.  Reference: configs/interm_{117m,10b}.yaml:27-45, res_slimvit.py:312-338."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

CONST = ["land_sea_mask", "orography", "lattitude", "landcover"]
ERA5_VARS = CONST + [
    "2m_temperature", "2m_temperature_max", "2m_temperature_min", "temperature_200", "temperature_500",
    "temperature_850", "10m_u_component_of_wind", "u_component_of_wind_200", "u_component_of_wind_500",
    "u_component_of_wind_850", "10m_v_component_of_wind", "v_component_of_wind_200", "v_component_of_wind_500",
    "v_component_of_wind_850", "specific_humidity_200", "specific_humidity_500", "specific_humidity_850",
    "total_precipitation_24hr", "volumetric_soil_water_layer_1"]
OUT_VARS = ["total_precipitation_24hr", "2m_temperature_min", "2m_temperature_max"]
VW = {"total_precipitation_24hr": 1.0, "2m_temperature_min": 10.0, "2m_temperature_max": 10.0}


def nerr(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def rel_l2(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_interm_117m_whole_model_forward_backward_vs_oracle():
    """BASELINE configs[1] at its real architecture and grid (2 of the 8 samples of the YAML batch: the oracle runs them
    on the CPU in seconds): prediction <= 2e-2, loss <= 1e-2, every parameter gradient within max(2e-2, 1.5 x the bf16 spread of
    that tensor, capped at 0.15) in normalised max error AND relative L2 -- the spread being the REFERENCE model's own bf16-vs-fp32
    movement on this very case (weights, inputs), committed as tests/golden/bf16_spread_configs.npz entry "interm_117m" by
    tests/golden/make_golden_bf16_spread.py, which also pins the oracle to the reference at this architecture (prediction 8e-8,
    gradients <= 1.5e-6).  Nothing about the tolerance is measured at test time."""
    from climate_learn.metrics import Bayesian_TV
    from climate_learn.trainer import training_step
    from oracle.harness import PINNED_CASES, build_pair
    assert PINNED_CASES["interm_117m"]["in_vars"] == ERA5_VARS and PINNED_CASES["interm_117m"]["out_vars"] == OUT_VARS
    model, sd, cfg, O, x, y, in_vars, out_vars = build_pair(**PINNED_CASES["interm_117m"])
    n = sum(p.numel() for p in model.parameters())
    assert 1.09e8 < n < 1.12e8 and len(in_vars) == 23            # 110 035 331 in the reference
    dev = torch.device("cuda")
    model = model.to(dev).eval()
    loss = training_step((x, y, in_vars, out_vars), 0, model, dev, VW, Bayesian_TV(aggregate_only=True))
    loss.backward()
    with torch.no_grad():
        pred = model(x.to(dev), in_vars, out_vars)
    torch.cuda.synchronize()
    sdo = {k: v.clone().requires_grad_() for k, v in sd.items()}
    pref = O.forward(sdo, cfg, x, in_vars, out_vars)
    assert tuple(pred.shape) == (2, 3, 128, 256) and nerr(pred, pref) < 2e-2
    ref = O.training_loss(sdo, cfg, x, y, in_vars, out_vars, "bayesian_tv", VW)
    ref.backward()
    assert abs(float(loss) - float(ref)) / abs(float(ref)) < 1e-2
    worst, l2 = {}, {}
    for name, p in model.named_parameters():
        g = sdo[name].grad
        if g is None:
            continue
        assert p.grad is not None, name
        worst[name], l2[name] = nerr(p.grad, g), rel_l2(p.grad, g)
    from oracle.harness import grad_tolerance, reference_spread
    sp = reference_spread("interm_117m", sd, x, y)
    assert nerr(pred, pref) < max(2e-2, 1.5 * sp["pred"])
    print(sorted(((round(e, 4), round(sp[k], 4), k) for k, e in worst.items()), reverse=True)[:10])
    bad = {k: (e, sp[k], l2[k], sp["l2." + k]) for k, e in worst.items()
           if e > grad_tolerance(sp[k], k) or l2[k] > grad_tolerance(sp["l2." + k])}
    assert len(worst) > 120 and not bad, bad
    from oracle.harness import admitted
    for k, e in worst.items():                                            # what passed above the 2e-2 floor goes on record
        admitted("interm_117m_vs_oracle", k, e, sp[k], grad_tolerance(sp[k]))
        admitted("interm_117m_vs_oracle", k, l2[k], sp["l2." + k], grad_tolerance(sp["l2." + k]), kind="l2")
    # beside the per-tensor bounds, the whole gradient: relative L2 over all parameters together
    num = sum(float((p.grad.detach().float().cpu().double() - sdo[n].grad.double()).pow(2).sum()) for n, p in model.named_parameters()
              if sdo[n].grad is not None)
    den = sum(float(sdo[n].grad.double().pow(2).sum()) for n, p in model.named_parameters() if sdo[n].grad is not None)
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5


def test_interm_10b_training_step_with_recompute():
    """BASELINE configs[3]: interm_10b (9.47 B parameters at the 128x256 grid) on ONE GPU, batch 1, every Block replayed
    in backward, bf16 compute / fp32 master, dropout on.  Size-independent properties: finite descending-capable step
    (finite loss, finite non-zero gradients in every unit, parameters move), recompute vs saved activations bit-identical
    on one Block's gradients, eval predictions independent of the batch neighbour."""
    import climate_learn as cl
    from climate_learn import _ops
    from climate_learn.metrics import Bayesian_TV
    from climate_learn.models.hub import Res_Slim_ViT
    from climate_learn.models.hub.components.vit_blocks import Block
    from climate_learn.trainer import training_step
    dev = torch.device("cuda")
    h, w = 128, 256
    with torch.device(dev):
        model = Res_Slim_ViT(ERA5_VARS, (h, w), 23, 3, 1, superres_mag=4, cnn_ratio=4, patch_size=2, drop_path=0.1,
                             drop_rate=0.1, learn_pos_emb=True, embed_dim=8192, depth=11, decoder_depth=4, num_heads=32,
                             mlp_ratio=4, FusedAttn_option=cl.FusedAttn.HIP)
    model.data_config(156.0, (h, w), 23, 3)
    n = sum(p.numel() for p in model.parameters())
    assert 9.4e9 < n < 9.55e9
    for blk in model.blocks:
        blk.recompute = True
    eng = cl.HipDataParallel(model, unit_types=(Block, nn.Sequential))
    opt = cl.load_optimizer(eng, "adamw", {"lr": 5e-4, "betas": (0.9, 0.99), "weight_decay": 1e-5})
    scaler = cl.HipGradScaler(init_scale=8192.0, growth_interval=100, min_scale=128.0)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 23, h, w, generator=g)
    y = torch.randn(1, 3, 721, 1440, generator=g)
    y[:, 0] = torch.log1p(torch.relu(y[:, 0]))
    batch = (x.to(dev), y.to(dev), ERA5_VARS, OUT_VARS)
    loss_fn = Bayesian_TV(aggregate_only=True)
    eng.train()
    cl.manual_seed(0)
    w0 = model.blocks[5].mlp.fc1.weight.detach()[:4, :64].clone()
    mark = _ops.seeds.mark()
    loss = training_step(batch, 0, eng, dev, VW, loss_fn)
    opt.zero_grad()
    scaler.scale(loss).backward()
    eng.finish_grad_sync()
    assert torch.isfinite(loss) and 0.0 < float(loss) < 1e3
    for bk in eng.buckets:                                  # every unit got finite, non-zero gradients
        for v in bk.grad_views:
            assert torch.isfinite(v.float()).all() and float(v.float().abs().sum()) > 0, bk.name
    g_rec = model.blocks[10].attn.qkv.weight._o2g.clone()
    g_rec2 = model.blocks[0].mlp.fc2.weight._o2g.clone()
    # the same step with saved activations on the first and last Block: bit-identical gradients there
    model.blocks[10].recompute = False
    model.blocks[0].recompute = False
    _ops.seeds.reset(mark)
    loss2 = training_step(batch, 0, eng, dev, VW, loss_fn)
    opt.zero_grad()
    scaler.scale(loss2).backward()
    eng.finish_grad_sync()
    assert float(loss2) == float(loss)
    assert torch.equal(model.blocks[10].attn.qkv.weight._o2g, g_rec)
    assert torch.equal(model.blocks[0].mlp.fc2.weight._o2g, g_rec2)
    scaler.step(opt)
    assert scaler.update() is False                         # no overflow at the initial loss scale
    torch.cuda.synchronize()
    assert not torch.equal(model.blocks[5].mlp.fc1.weight.detach()[:4, :64], w0)      # the optimizer moved the weights
    del g_rec, g_rec2
    # batch independence of the eval forward at the real size
    eng.eval()
    opt.zero_grad()
    x2 = torch.randn(2, 23, h, w, generator=g).to(dev)
    with torch.no_grad():
        y2 = eng(x2, ERA5_VARS, OUT_VARS)
        y1 = eng(x2[1:2].contiguous(), ERA5_VARS, OUT_VARS)
    assert y2.shape == (2, 3, 512, 1024) and torch.isfinite(y2).all() and torch.equal(y2[1:2], y1)


def test_interm_1b_daymet_like_hybrid_perceptual_step(monkeypatch):
    """BASELINE configs[4] AT ITS SIZE: interm_1b (D3072 / 24 heads of 128 / depth 8), 7 Daymet-like inputs ->
    3 outputs, the 96x192 -> 384x768 tile (L = 4608), hybrid loss `perceptual_lat_mse` = reference `perceptual`
    (metrics.py:119-187, functional.py:17-33) + intended `lat_mse` (metrics.py:295-316), var weights {1,10,10}
    (configs/interm_1b.yaml:225-232), train mode, loss-scaled AdamW.  The whole oracle cannot run at this width; checked:
    finite step with finite non-zero gradients in every unit, seed determinism (same seeds -> same loss and gradients bit for
    bit, new seeds -> different), the loss's LPIPS + L1 + lat-MSE value of the step's own predictions against the CPU oracle
    (the VGG stack fits the CPU at 384x768), and batch independence of the eval prediction."""
    import climate_learn as cl
    from climate_learn import _ops
    from climate_learn.metrics.lpips_hip import LPIPSVGG16
    from climate_learn.metrics.utils import MetricsMetaInfo
    from climate_learn.models.hub import Res_Slim_ViT
    from climate_learn.models.hub.components.vit_blocks import Block
    from climate_learn.trainer import clip_replace_constant, training_step
    from oracle import orbit2_oracle as O
    monkeypatch.delenv("ORBIT2_LPIPS_WEIGHTS", raising=False)
    monkeypatch.setenv("ORBIT2_LPIPS_SYNTHETIC", "1")
    dev = torch.device("cuda")
    h, w, B = 96, 192, 2
    in_vars = CONST + OUT_VARS
    with torch.device(dev):
        model = Res_Slim_ViT(in_vars, (h, w), 7, 3, 1, superres_mag=4, cnn_ratio=4, patch_size=2, drop_path=0.1,
                             drop_rate=0.1, learn_pos_emb=True, embed_dim=3072, depth=8, decoder_depth=4, num_heads=24,
                             mlp_ratio=4, FusedAttn_option=cl.FusedAttn.HIP)
    model.data_config(16.0, (h, w), 7, 3)
    eng = cl.HipDataParallel(model, unit_types=(Block, nn.Sequential))
    opt = cl.load_optimizer(eng, "adamw", {"lr": 5e-4, "betas": (0.9, 0.99), "weight_decay": 1e-5})
    scaler = cl.HipGradScaler(init_scale=8192.0, growth_interval=100, min_scale=128.0)
    lat = np.linspace(25.0, 50.0, 4 * h)
    loss_fn = cl.load_loss(dev, None, "perceptual_lat_mse", True, MetricsMetaInfo(in_vars, OUT_VARS, lat, None, None))
    sd_l = {k: (v.to(torch.bfloat16).float() if "lin" not in k else v) for k, v in O.init_lpips_weights(5).items()}
    loss_fn.loss_fn = LPIPSVGG16(dev, sd_l)                  # the same stand-in LPIPS weights as the oracle below
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, 7, h, w, generator=g)
    y = torch.randn(B, 3, 4 * h, 4 * w, generator=g)
    y[:, 0] = torch.log1p(torch.relu(y[:, 0]))
    batch = (x.to(dev), y.to(dev), in_vars, OUT_VARS)
    eng.train()
    cl.manual_seed(0)

    def one(mark=None):
        if mark is not None:
            _ops.seeds.reset(mark)
        loss = training_step(batch, 0, eng, dev, VW, loss_fn)
        opt.zero_grad()
        scaler.scale(loss).backward()
        eng.finish_grad_sync()
        return loss

    mark = _ops.seeds.mark()
    l1 = one()
    assert torch.isfinite(l1) and 0.0 < float(l1) < 1e3
    for bk in eng.buckets:
        for v in bk.grad_views:
            assert torch.isfinite(v.float()).all() and float(v.float().abs().sum()) > 0, bk.name
    g1 = model.blocks[7].attn.qkv.weight._o2g.clone()
    gh = model.head[8].weight._o2g.clone()
    l2 = one(mark)           # same seeds: the same masks -> the same step, bit for bit (no float atomics anywhere in it)
    assert float(l2) == float(l1)
    assert torch.equal(model.blocks[7].attn.qkv.weight._o2g, g1) and torch.equal(model.head[8].weight._o2g, gh)
    l3 = one()                                               # fresh dropout / DropPath masks: a different step
    assert abs(float(l3) - float(l1)) > 1e-5 * abs(float(l1)) and rel_l2(model.blocks[7].attn.qkv.weight._o2g, g1) > 0.05
    scaler.step(opt)
    assert scaler.update() is False
    # the loss object on the step's own predictions against the oracle (CPU, fp32): L1 + 0.5 LPIPS + lat-weighted MSE
    eng.eval()
    with torch.no_grad():
        pred = eng(batch[0], in_vars, OUT_VARS)
        raw = pred.clone()                                      # clip_replace_constant clamps precipitation IN PLACE
        yhat = clip_replace_constant(batch[1], pred, OUT_VARS)
        val = loss_fn(yhat, batch[1], var_names=OUT_VARS, var_weights=VW)
        p1 = eng(batch[0][1:2].contiguous(), in_vars, OUT_VARS)
    assert raw.shape == (B, 3, 4 * h, 4 * w) and torch.equal(raw[1:2], p1)          # batch independence
    yc, tc = yhat.float().cpu(), batch[1].float().cpu()
    lp = float(O.lpips_vgg(yc, tc, sd_l).mean())
    ref = float((yc - tc).abs().mean()) + 0.5 * lp + float(O.mse(yc, tc, OUT_VARS, VW, True, O.lat_weights(lat, 4 * h)))
    assert abs(float(val) - ref) / ref < 1e-2, (float(val), ref)
    per = float(loss_fn.loss_fn.perceptual(yhat, batch[1]))
    lp_hip = 2.0 * (per - float((yc - tc).abs().mean()))
    assert abs(lp_hip - lp) / lp < 3e-2, (lp_hip, lp)          # the LPIPS term alone (13 bf16 layers deep)
