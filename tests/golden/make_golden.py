#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE's own hot-path modules on CPU and
stores inputs / weights / outputs / gradients as small .npz fixtures.

Runs ONLY in the build container (where /root/reference exists). Nothing under
tests/, bench.py or smoke() imports this module; they read the .npz files it wrote.
No reference source text is stored -- only numeric inputs and outputs.

Recipe: the reference package cannot be imported whole (torchvision,
xformers, timm, lpips ... are absent), so empty namespace packages whose __path__ points
into /root/reference/src/climate_learn are registered and the third-party symbols the hot
path touches are stubbed with their documented behaviour:
  timm trunc_normal_ -> torch.nn.init.trunc_normal_, DropPath -> per-sample Bernoulli,
  to_2tuple, _assert, GlobalResponseNorm (unused), xformers (unused: FusedAttn NONE/DEFAULT).

Usage:  python tests/golden/make_golden.py          (writes tests/golden/*.npz)
"""
import importlib
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference/src/climate_learn"
OUT = os.path.dirname(os.path.abspath(__file__))


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def install_shims():
    _pkg("climate_learn", REF)
    _pkg("climate_learn.utils", REF + "/utils")
    _pkg("climate_learn.models", REF + "/models")
    _pkg("climate_learn.models.hub", REF + "/models/hub")
    _pkg("climate_learn.models.hub.components", REF + "/models/hub/components")
    _pkg("climate_learn.metrics", REF + "/metrics")

    # --- timm -------------------------------------------------------------
    timm = _pkg("timm")
    tm = _pkg("timm.models")
    vt = _pkg("timm.models.vision_transformer")
    vt.trunc_normal_ = torch.nn.init.trunc_normal_
    tl = _pkg("timm.layers")

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob = drop_prob
            self.scale_by_keep = scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            mask = x.new_empty(shape).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                mask.div_(keep)
            return x * mask

    tl.DropPath = DropPath
    hl = _pkg("timm.layers.helpers")
    hl.to_2tuple = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    tu = _pkg("timm.layers.trace_utils")

    def _assert(c, m):
        assert c, m

    tu._assert = _assert
    grn = _pkg("timm.layers.grn")
    grn.GlobalResponseNorm = nn.Identity
    # --- xformers (CK path not used for goldens) ----------------------------
    xf = _pkg("xformers")
    _pkg("xformers.components")
    _pkg("xformers.components.attention")
    core = _pkg("xformers.components.attention.core")
    core.scaled_dot_product_attention = None
    # --- loss-side third parties (never called for mse / bayesian_tv) -------
    lp = _pkg("lpips")
    lp.LPIPS = None
    lp.NetLinLayer = None
    tv = _pkg("torchvision")
    tvm = _pkg("torchvision.models")
    tvm.vgg16 = None
    tmx = _pkg("torchmetrics")
    _pkg("torchmetrics.functional")
    tmi = _pkg("torchmetrics.functional.image")
    tmi.image_gradients = None


def t2n(t):
    return t.detach().cpu().numpy().astype(np.float32) if t.dtype.is_floating_point else t.detach().cpu().numpy()


def randomize_(module, gen, scale=0.05):
    """Make every parameter non-degenerate (var_embed/var_query/biases are zero at init)."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.ndim == 1 and ("norm" in n and n.endswith("weight")):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gen))
            elif n == "pos_embed":
                p.add_(0.02 * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(scale * torch.randn(p.shape, generator=gen) * (4.0 if p.ndim <= 1 else 1.0)
                        if p.numel() > 0 else p)


def main():
    install_shims()
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)

    rs = importlib.import_module("climate_learn.models.hub.res_slimvit")
    att = importlib.import_module("climate_learn.models.hub.components.attention")
    mlp = importlib.import_module("climate_learn.models.hub.components.mlp")
    blk = importlib.import_module("climate_learn.models.hub.components.vit_blocks")
    pe = importlib.import_module("climate_learn.models.hub.components.patch_embed")
    pos = importlib.import_module("climate_learn.models.hub.components.pos_embed")
    fa = importlib.import_module("climate_learn.utils.fused_attn")
    fn = importlib.import_module("climate_learn.metrics.functional")
    sch = importlib.import_module("climate_learn.models.lr_scheduler")
    FusedAttn = fa.FusedAttn

    # ------------------------------------------------------------------ components
    g = torch.Generator().manual_seed(1234)
    comp = {}
    D, Hd, V, B = 64, 4, 5, 2
    gh, gw = 8, 16          # input grid
    p = 2
    L = gh * gw // p // p   # 32 tokens

    # PatchEmbed (one variable)
    m = pe.PatchEmbed((gh, gw), p, 1, D)
    randomize_(m, g, 0.3)
    x = torch.randn(B, 1, gh, gw, generator=g, requires_grad=True)
    y = m(x)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    comp.update({"pe.x": t2n(x), "pe.w": t2n(m.proj.weight), "pe.b": t2n(m.proj.bias), "pe.y": t2n(y),
                 "pe.go": t2n(go), "pe.gx": t2n(x.grad), "pe.gw": t2n(m.proj.weight.grad),
                 "pe.gb": t2n(m.proj.bias.grad)})

    # VariableMapping_Attention, NONE and DEFAULT must agree
    for mode in ("NONE", "DEFAULT"):
        torch.manual_seed(7)
        m = att.VariableMapping_Attention(D, fused_attn=FusedAttn[mode], num_heads=Hd, qkv_bias=False)
        gg = torch.Generator().manual_seed(99)
        randomize_(m, gg, 0.2)
        xin = torch.randn(B * L, V, D, generator=gg, requires_grad=True)
        vq = torch.randn(1, 1, D, generator=gg, requires_grad=True)
        y = m(vq.expand(B * L, -1, -1).contiguous(), xin)
        go = torch.randn(y.shape, generator=gg)
        y.backward(go)
        if mode == "NONE":
            comp.update({"va.x": t2n(xin), "va.vq": t2n(vq), "va.wq": t2n(m.q.weight), "va.wkv": t2n(m.kv.weight),
                         "va.wp": t2n(m.proj.weight), "va.bp": t2n(m.proj.bias), "va.y": t2n(y), "va.go": t2n(go),
                         "va.gx": t2n(xin.grad), "va.gvq": t2n(vq.grad), "va.gwq": t2n(m.q.weight.grad),
                         "va.gwkv": t2n(m.kv.weight.grad), "va.gwp": t2n(m.proj.weight.grad),
                         "va.gbp": t2n(m.proj.bias.grad)})
        else:
            comp["va.y_default"] = t2n(y)

    # Attention
    m = att.Attention(D, fused_attn=FusedAttn.NONE, num_heads=Hd, qkv_bias=True)
    randomize_(m, g, 0.2)
    m.eval()
    x = torch.randn(B, L, D, generator=g, requires_grad=True)
    y = m(x)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    comp.update({"at.x": t2n(x), "at.wqkv": t2n(m.qkv.weight), "at.bqkv": t2n(m.qkv.bias),
                 "at.wp": t2n(m.proj.weight), "at.bp": t2n(m.proj.bias), "at.y": t2n(y), "at.go": t2n(go),
                 "at.gx": t2n(x.grad), "at.gwqkv": t2n(m.qkv.weight.grad), "at.gbqkv": t2n(m.qkv.bias.grad),
                 "at.gwp": t2n(m.proj.weight.grad), "at.gbp": t2n(m.proj.bias.grad)})
    m.fused_attn = FusedAttn.DEFAULT
    comp["at.y_default"] = t2n(m(x))

    # Mlp
    m = mlp.Mlp(D, 4 * D, drop=0.0)
    randomize_(m, g, 0.2)
    x = torch.randn(B, L, D, generator=g, requires_grad=True)
    y = m(x)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    comp.update({"ml.x": t2n(x), "ml.w1": t2n(m.fc1.weight), "ml.b1": t2n(m.fc1.bias), "ml.w2": t2n(m.fc2.weight),
                 "ml.b2": t2n(m.fc2.bias), "ml.y": t2n(y), "ml.go": t2n(go), "ml.gx": t2n(x.grad),
                 "ml.gw1": t2n(m.fc1.weight.grad), "ml.gb1": t2n(m.fc1.bias.grad),
                 "ml.gw2": t2n(m.fc2.weight.grad), "ml.gb2": t2n(m.fc2.bias.grad)})

    # Block (eval -> no dropout / droppath)
    m = blk.Block(D, Hd, fused_attn=FusedAttn.NONE, mlp_ratio=4.0, qkv_bias=True, drop_path=0.1,
                  proj_drop=0.1, attn_drop=0.1)
    randomize_(m, g, 0.2)
    m.eval()
    x = torch.randn(B, L, D, generator=g, requires_grad=True)
    y = m(x)
    go = torch.randn(y.shape, generator=g)
    y.backward(go)
    comp.update({"bk.x": t2n(x), "bk.y": t2n(y), "bk.go": t2n(go), "bk.gx": t2n(x.grad)})
    for n, prm in m.named_parameters():
        comp["bk.p." + n] = t2n(prm)
        comp["bk.g." + n] = t2n(prm.grad)

    # sincos pos-embed + both branches of the on-the-fly interpolation
    comp["pos.sincos_4x8_64"] = pos.get_2d_sincos_pos_embed(64, 4, 8).astype(np.float32)
    comp["pos.sincos_16x32_256"] = pos.get_2d_sincos_pos_embed(256, 16, 32).astype(np.float32)
    pemb = torch.randn(1, 4 * 8, D, generator=g)
    comp["pos.in"] = t2n(pemb)
    comp["pos.same"] = t2n(pos.interpolate_pos_embed_on_the_fly(pemb, p, (8, 16)))
    comp["pos.up_12x24"] = t2n(pos.interpolate_pos_embed_on_the_fly(pemb, p, (24, 48)))
    comp["pos.down_2x4"] = t2n(pos.interpolate_pos_embed_on_the_fly(pemb, p, (4, 8)))
    np.savez_compressed(os.path.join(OUT, "components_tiny.npz"), **comp)

    # ------------------------------------------------------------------ whole model
    CONST = ["land_sea_mask", "orography", "lattitude", "landcover"]
    cases = {
        "v5c1": dict(in_vars=CONST + ["total_precipitation_24hr"], out_vars=["total_precipitation_24hr"],
                     grid=(16, 32), run_grid=(16, 32), D=64, depth=2, heads=4, dd=2),
        "v7c3": dict(in_vars=CONST + ["total_precipitation_24hr", "2m_temperature_min", "2m_temperature_max"],
                     out_vars=["total_precipitation_24hr", "2m_temperature_min", "2m_temperature_max"],
                     grid=(16, 32), run_grid=(16, 32), D=64, depth=2, heads=2, dd=1),
        # shapes the HIP kernels support (head dim 64, L = 128 tokens): the GPU whole-model parity fixture
        "v5c1_hd64": dict(in_vars=CONST + ["total_precipitation_24hr"], out_vars=["total_precipitation_24hr"],
                          grid=(16, 32), run_grid=(16, 32), D=128, depth=2, heads=2, dd=1),
        "v7c3_hd64": dict(in_vars=["2m_temperature_max", "lattitude", "total_precipitation_24hr", "orography",
                                   "landcover", "land_sea_mask", "2m_temperature_min"],
                          out_vars=["2m_temperature_min", "total_precipitation_24hr", "2m_temperature_max"],
                          default_vars=CONST + ["2m_temperature", "total_precipitation_24hr", "2m_temperature_min",
                                                "2m_temperature_max"],
                          grid=(16, 32), run_grid=(16, 32), D=128, depth=1, heads=2, dd=2),
        # data_config'd to a bigger grid than the init grid -> bicubic pos-embed branch; default_vars
        # is a superset of in_vars and in another order -> exercises the var-id gather
        "v6c2_regrid": dict(in_vars=["2m_temperature", "lattitude", "orography", "landcover", "land_sea_mask",
                                     "total_precipitation_24hr"],
                            out_vars=["total_precipitation_24hr", "2m_temperature"],
                            default_vars=CONST + ["2m_temperature", "10m_u_component_of_wind",
                                                  "total_precipitation_24hr"],
                            grid=(8, 16), run_grid=(16, 32), D=32, depth=1, heads=2, dd=1),
        # the same re-gridded case at a head dim the HIP attention supports (64): GPU parity of the bicubic branch
        "v6c2_regrid_hd64": dict(in_vars=["2m_temperature", "lattitude", "orography", "landcover", "land_sea_mask",
                                          "total_precipitation_24hr"],
                                 out_vars=["total_precipitation_24hr", "2m_temperature"],
                                 default_vars=CONST + ["2m_temperature", "10m_u_component_of_wind",
                                                       "total_precipitation_24hr"],
                                 grid=(8, 16), run_grid=(16, 32), D=128, depth=1, heads=2, dd=1),
    }
    for tag, c in cases.items():
        torch.manual_seed(0)
        dv = c.get("default_vars", c["in_vars"])
        Vn, C = len(c["in_vars"]), len(c["out_vars"])
        model = rs.Res_Slim_ViT(dv, c["grid"], Vn, C, history=1, superres_mag=4, cnn_ratio=4, patch_size=2,
                                drop_path=0.1, drop_rate=0.1, learn_pos_emb=True, embed_dim=c["D"],
                                depth=c["depth"], decoder_depth=c["dd"], num_heads=c["heads"], mlp_ratio=4,
                                FusedAttn_option=FusedAttn.NONE)
        gg = torch.Generator().manual_seed(4321)
        randomize_(model, gg, 0.08)
        model.data_config(156.0, c["run_grid"], Vn, C)
        model.eval()
        h, w = c["run_grid"]
        x = torch.randn(B, Vn, h, w, generator=gg)
        y = torch.randn(B, C, 4 * h + 3, 4 * w + 5, generator=gg)      # bigger than 4x -> cropped by the step
        pi = c["out_vars"].index("total_precipitation_24hr")
        y[:, pi] = torch.log1p(torch.relu(y[:, pi]))
        out = {"x": t2n(x), "y": t2n(y)}
        pred = model(x, c["in_vars"], c["out_vars"])
        out["pred"] = t2n(pred)
        # training_step semantics (examples/intermediate_downscaling.py:267-299): clamp precip, crop target
        yhat = pred.clone()
        yhat[:, pi] = torch.clamp(pred[:, pi], min=0.0)
        yc = y[:, :, : yhat.shape[2], : yhat.shape[3]]
        vw = {"total_precipitation_24hr": 1.0, "2m_temperature_min": 10.0, "2m_temperature_max": 10.0,
              "2m_temperature": 10.0}
        lat = np.linspace(-88.0, 88.0, yhat.shape[2])
        wl = np.cos(np.deg2rad(lat))
        wl = torch.from_numpy(wl / wl.mean()).view(1, 1, -1, 1).float()
        for lname, lfn, kw in (("mse", fn.mse, {}), ("bayesian_tv", fn.bayesian_tv, {}),
                               ("lat_mse", fn.mse, {"lat_weights": wl})):
            model.zero_grad()
            full = lfn(yhat, yc, c["out_vars"], vw, False, **kw)
            out["loss." + lname] = t2n(full)
            if lname != "lat_mse":
                full[-1].backward(retain_graph=True)
                for n, prm in model.named_parameters():
                    if prm.grad is not None:
                        out["g.%s.%s" % (lname, n)] = t2n(prm.grad)
        out["lat"] = lat.astype(np.float64)
        for n, prm in model.state_dict().items():
            out["p." + n] = t2n(prm)
        # three AdamW steps, fp32, dropout off (loss trajectory pins optimizer + scheduler semantics)
        opt = torch.optim.AdamW(model.parameters(), lr=5e-4, betas=(0.9, 0.99), weight_decay=1e-5)
        traj = []
        for _ in range(3):
            pr = model(x, c["in_vars"], c["out_vars"])
            yh = pr.clone()
            yh[:, pi] = torch.clamp(pr[:, pi], min=0.0)
            ls = fn.bayesian_tv(yh, yc, c["out_vars"], vw, True)
            traj.append(float(ls))
            opt.zero_grad()
            ls.backward()
            opt.step()
        out["adamw.loss_traj"] = np.array(traj, dtype=np.float64)
        out["adamw.p_after.head.0.weight"] = t2n(model.state_dict()["head.0.weight"])
        out["adamw.p_after.var_query"] = t2n(model.state_dict()["var_query"])
        np.savez_compressed(os.path.join(OUT, "model_%s.npz" % tag), **out)

    # ------------------------------------------------------------------ losses on raw tensors
    gl = torch.Generator().manual_seed(77)
    pr = torch.randn(2, 3, 16, 32, generator=gl, requires_grad=True)
    tg = torch.randn(2, 3, 16, 32, generator=gl)
    names = ["total_precipitation_24hr", "2m_temperature_min", "2m_temperature_max"]
    vw = {"2m_temperature_min": 10.0, "2m_temperature_max": 10.0, "total_precipitation_24hr": 1.0}
    lat = np.linspace(-90, 90, 16)
    wl = np.cos(np.deg2rad(lat))
    wl = torch.from_numpy(wl / wl.mean()).view(1, 1, -1, 1).float()
    ls = {"pred": t2n(pr), "target": t2n(tg), "lat": lat}
    for lname, lfn, kw in (("mse", fn.mse, {}), ("bayesian_tv", fn.bayesian_tv, {}),
                           ("lat_mse", fn.mse, {"lat_weights": wl}),
                           ("lat_bayesian_tv", fn.bayesian_tv, {"lat_weights": wl})):
        pr.grad = None
        full = lfn(pr, tg, names, vw, False, **kw)
        full[-1].backward()
        ls[lname] = t2n(full)
        ls[lname + ".gpred"] = t2n(pr.grad)
        ls[lname + ".noweights"] = t2n(lfn(pr, tg, None, None, False, **kw))
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **ls)

    # ------------------------------------------------------------------ LR schedule
    prm = [nn.Parameter(torch.zeros(1))]
    opt = torch.optim.AdamW(prm, lr=5e-4)
    sc = sch.LinearWarmupCosineAnnealingLR(opt, warmup_epochs=2, max_epochs=100, warmup_start_lr=1e-7, eta_min=1e-8)
    lrs = []
    for _ in range(100):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sc.step()
    opt2 = torch.optim.AdamW(prm, lr=2e-4)
    sc2 = sch.LinearWarmupCosineAnnealingLR(opt2, warmup_epochs=5, max_epochs=30, warmup_start_lr=1e-6, eta_min=1e-7)
    lrs2 = []
    for _ in range(40):
        lrs2.append(opt2.param_groups[0]["lr"])
        opt2.step()
        sc2.step()
    np.savez_compressed(os.path.join(OUT, "lr_schedule.npz"), lr_w2_m100=np.array(lrs), lr_w5_m30=np.array(lrs2))

    # ------------------------------------------------------------------ NpyReader tiling / sharding (data plane)
    spec = importlib.util.spec_from_file_location("ref_iterdataset", REF + "/data/iterdataset.py")
    rid = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rid)
    tg = {}
    with tempfile.TemporaryDirectory() as td:
        files_in, files_out = [], []
        for f in range(3):
            yy, xx = np.meshgrid(np.arange(16), np.arange(32), indexing="ij")
            lo = (f * 1e6 + yy * 1000 + xx).astype(np.float64)[None, None].repeat(2, 0)          # [T=2,1,16,32]
            YY, XX = np.meshgrid(np.arange(64), np.arange(128), indexing="ij")
            hi = (f * 1e6 + YY * 1000 + XX).astype(np.float64)[None, None].repeat(2, 0)          # [2,1,64,128]
            pi, po = os.path.join(td, "in_%d.npz" % f), os.path.join(td, "out_%d.npz" % f)
            np.savez(pi, a=lo, b=lo + 0.5)
            np.savez(po, c=hi)
            files_in.append(pi)
            files_out.append(po)
        for div, ov in ((1, 0), (2, 2), (4, 3), (2, 1)):
            rd = rid.NpyReader(files_in, files_out, ["a", "b"], ["c"], data_par_size=1, div=div, overlap=ov)
            rec = []
            for xin, yout, _, _ in rd:
                a, c = xin["a"], yout["c"]
                rec.append([a.shape[1], a.shape[2], a[0, 0, 0], a[0, -1, -1], c.shape[1], c.shape[2], c[0, 0, 0], c[0, -1, -1]])
            tg["tiles_div%d_ov%d" % (div, ov)] = np.array(rec, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "tiling.npz"), **tg)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print("  %-28s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == "__main__":
    main()
