"""GPU tests of the forward-only tiled inference path and the evaluation metrics."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_eval_metrics_match_reference_golden(golden_dir):
    from climate_learn.metrics import functional as fn
    from climate_learn.metrics import METRICS_REGISTRY, MetricsMetaInfo
    z = np.load(os.path.join(golden_dir, "eval_metrics.npz"))
    pred, target = torch.from_numpy(z["pred"]).cuda(), torch.from_numpy(z["target"]).cuda()
    for name, got in (("rmse", fn.rmse(pred, target)), ("pearson", fn.pearson(pred, target)),
                      ("mean_bias", fn.mean_bias(pred, target))):
        assert np.allclose(got.cpu().numpy(), z[name], rtol=2e-5, atol=2e-6), name
    assert np.allclose(float(fn.rmse(pred, target, True)), z["rmse.agg"], rtol=2e-5)
    meta = MetricsMetaInfo(["a", "b", "c"], ["a", "b", "c"], z["lat"], np.zeros(40), None)
    lat_rmse = METRICS_REGISTRY["lat_rmse"](aggregate_only=False, metainfo=meta)
    assert np.allclose(lat_rmse(pred, target).cpu().numpy(), z["lat_rmse"], rtol=2e-5)
    for n in ("rmse", "pearson", "mean_bias"):
        m = METRICS_REGISTRY[n](aggregate_only=True, metainfo=meta)
        assert np.allclose(float(m(pred, target)), z[n + ".agg"], rtol=2e-5, atol=2e-6)
    # mae / anomaly correlation (forecasting-side metrics of the registry), climatology [C,H,W]
    clim = torch.from_numpy(z["clim"])
    meta_c = MetricsMetaInfo(["a", "b", "c"], ["a", "b", "c"], z["lat"], np.zeros(40), clim)
    assert np.allclose(METRICS_REGISTRY["mae"](aggregate_only=False, metainfo=meta_c)(pred, target).cpu().numpy(), z["mae"], rtol=2e-5)
    assert np.allclose(fn.mae(pred, target, False, torch.from_numpy(np.cos(np.deg2rad(z["lat"])) / np.cos(np.deg2rad(z["lat"])).mean()).float().view(1, 1, -1, 1)).cpu().numpy(),
                       z["lat_mae"], rtol=2e-5)
    la = METRICS_REGISTRY["lat_acc"](aggregate_only=False, metainfo=meta_c)
    assert np.allclose(la(pred, target).cpu().numpy(), z["lat_acc"], rtol=5e-5, atol=5e-6)
    ua = METRICS_REGISTRY["acc"](aggregate_only=False, metainfo=meta_c)
    assert np.allclose(ua(pred, target).cpu().numpy(), z["acc_unit_weights"], rtol=5e-5, atol=5e-6)
    # a target larger than the prediction is consumed through its top-left crop
    big = torch.zeros(3, 3, 30, 47, device="cuda")
    big[:, :, :24, :40] = target
    assert np.allclose(fn.rmse(pred, big).cpu().numpy(), z["rmse"], rtol=2e-5)


def test_tiled_predict_matches_per_tile_forward():
    from oracle.harness import build_pair
    from climate_learn.trainer import clip_replace_constant
    from climate_learn.utils.visualize import tiled_predict, tile_windows
    model, sd, cfg, O, x, y, in_vars, out_vars = build_pair(D=128, depth=1, heads=2, grid=(16, 32), B=1, seed=7)
    model = model.cuda().eval()
    g = torch.Generator().manual_seed(2)
    X = torch.randn(1, len(in_vars), 32, 64, generator=g).cuda()           # a field 2x the training tile
    Y = torch.randn(1, len(out_vars), 128, 256, generator=g).cuda()
    cfgd = (model.spatial_resolution, model.img_size, model.in_channels, model.out_channels)
    with torch.no_grad():
        model.data_config(cfgd[0], (32, 64), cfgd[2], cfgd[3])
        direct = clip_replace_constant(Y, model(X, in_vars, out_vars), out_vars)
        model.data_config(*cfgd)
    one = tiled_predict(model, X, Y, in_vars, out_vars, 1, 0)
    assert torch.equal(one, direct.float())
    div, ov = 2, 4
    st = tiled_predict(model, X, Y, in_vars, out_vars, div, ov)
    assert st.shape == (1, len(out_vars), 128, 256) and torch.isfinite(st).all()
    assert tuple(model.img_size) == tuple(cfgd[1])                # tiled_predict restored the caller's data_config
    for t in tile_windows(32, 64, 128, 256, div, ov):
        (yi1, yi2), (xi1, xi2) = t["inp"]
        (yo1, yo2), (xo1, xo2) = t["out"]
        with torch.no_grad():
            model.data_config(cfgd[0], (yi2 - yi1, xi2 - xi1), cfgd[2], cfgd[3])
            p = clip_replace_constant(Y[:, :, yo1:yo2, xo1:xo2],
                                      model(X[:, :, yi1:yi2, xi1:xi2].contiguous(), in_vars, out_vars), out_vars)
        (ya, yb), (xa, xb) = t["crop_out"]
        (ra, rb), (ca, cb) = t["place_out"]
        assert torch.equal(st[:, :, ra:rb, ca:cb], p[:, :, ya:yb, xa:xb].float())


def test_visualize_at_index_returns_stitched_north_up_fields(tmp_path):
    import climate_learn as cl
    from oracle.harness import build_pair
    from climate_learn.transforms import Denormalize
    from climate_learn.utils.visualize import visualize_at_index
    model, sd, cfg, O, x, y, in_vars, out_vars = build_pair(D=128, depth=1, heads=2, grid=(16, 32), B=1, seed=4)
    model = model.cuda().eval()
    dm = cl.data.IterDataModule("downscaling", "ERA5_lo", "ERA5_hi", in_vars, out_vars=out_vars, batch_size=1,
                                lowres_hw=(32, 64), highres_hw=(128, 256))
    dm.setup()
    den = Denormalize(dm)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        res = visualize_at_index(model, dm, dm, out_vars, den, den, "total_precipitation_24hr", "ERA5", "cuda", 2, 4)
    finally:
        os.chdir(cwd)
    assert res["inputs"].shape == (32, 64) and res["preds"].shape == (128, 256) and res["groundtruths"].shape == (128, 256)
    g64, p64 = res["groundtruths"].astype(np.float64), res["preds"].astype(np.float64)      # PSNR through orbit2_eval_moments
    assert abs(res["psnr"] - 10 * np.log10((g64.max() - g64.min()) ** 2 / ((g64 - p64) ** 2).mean())) < 1e-3 and -1.0 <= res["ssim"] <= 1.0
    xb, yb = dm.test_dataloader()[0][:2]
    assert np.array_equal(res["groundtruths"], np.flip(yb[0, 0].numpy(), 0))       # north-up flip, precip not denormalised
    assert np.array_equal(res["inputs"], np.flip(xb[0, in_vars.index("total_precipitation_24hr")].numpy(), 0))
