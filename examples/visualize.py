#!/usr/bin/env python3
"""Counterpart of the reference's examples/visualize.py (tiled inference + stitching, :340-478).

    python examples/visualize.py configs/inference.yaml

Builds the model from the YAML (same schema as the training configs), optionally loads `trainer.pretrain`
(shape-tolerant, like the reference), cuts test sample 0 into `tiling.div`^2 tiles with an `overlap` halo, runs the
HIP forward on every tile, stitches the interiors, and reports the denormalised rmse / pearson / mean_bias of the
stitched field (the reference's validation metrics, loaders.py:247-255).  Forward only; one process, one GPU."""
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd")]

import climate_learn as cl                                                        # noqa: E402
from climate_learn.utils.fused_attn import FusedAttn                              # noqa: E402
from climate_learn.utils.visualize import tiled_predict, visualize_at_index      # noqa: E402


def main():
    conf = yaml.load(open(sys.argv[1]), Loader=yaml.FullLoader)
    tr, mc, dc = conf["trainer"], conf["model"], conf["data"]
    tiling = conf.get("tiling", {}) or {}
    div, overlap = (tiling.get("div", 1), tiling.get("overlap", 0)) if tiling.get("do_tiling", False) else (1, 0)
    local_rank = int(os.environ.get("SLURM_LOCALID", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    data_key = next(iter(dc["low_res_dir"]))
    in_vars, out_vars = dc["dict_in_variables"][data_key], dc["dict_out_variables"][data_key]
    syn = (dc.get("synthetic") or {}).get(data_key, {})
    # the untiled module supplies whole fields (the reference's dm_vis: div=1, overlap=0)
    dm_vis = cl.data.IterDataModule("downscaling", dc["low_res_dir"][data_key], dc["high_res_dir"][data_key], in_vars,
                                    out_vars=out_vars, subsample=1, batch_size=1, div=1, overlap=0,
                                    lowres_hw=tuple(syn.get("lowres_hw", (32, 64))),
                                    highres_hw=tuple(syn["highres_hw"]) if "highres_hw" in syn else None,
                                    steps_per_epoch=syn.get("steps_per_epoch", 1)).to(device)
    dm_vis.setup()
    with torch.device(device):
        out = cl.load_downscaling_module(
            device, data_module=dm_vis, architecture=mc["preset"], train_loss=tr["train_loss"],
            model_kwargs={"default_vars": dc["default_vars"], "superres_mag": mc["superres_mag"],
                          "cnn_ratio": mc["cnn_ratio"], "patch_size": mc["patch_size"], "embed_dim": mc["embed_dim"],
                          "depth": mc["depth"], "decoder_depth": mc["decoder_depth"], "num_heads": mc["num_heads"],
                          "mlp_ratio": mc["mlp_ratio"], "drop_path": mc["drop_path"], "drop_rate": mc["drop_rate"],
                          "tensor_par_size": 1, "tensor_par_group": None, "FusedAttn_option": FusedAttn.CK})
    model, test_losses, test_transforms = out[0], out[3], out[6]
    if tr.get("pretrain"):
        print("load pretrained model", tr["pretrain"], flush=True)
        cl.utils.load_pretrained_weights(model, str(tr["pretrain"]), verbose=True)
    model = model.to(device).eval()
    model.data_config(dc["spatial_resolution"][data_key], model.img_size, len(in_vars), len(out_vars))
    denorm = test_transforms[0]
    variable = "total_precipitation_24hr" if "total_precipitation_24hr" in out_vars else out_vars[0]
    res = visualize_at_index(model, dm_vis, dm_vis, out_list=out_vars, in_transform=denorm, out_transform=denorm,
                             variable=variable, src=data_key, device=device, div=div, overlap=overlap, index=0)
    print("stitched", {k: (v.shape if hasattr(v, "shape") else v) for k, v in res.items()}, flush=True)
    x, y, iv, ov = next(iter(dm_vis.test_dataloader()))[:4]
    x, y = x.to(device), y.to(device)
    pred = denorm(tiled_predict(model, x, y, iv, ov, div, overlap))
    gt = denorm(y[:, :, : pred.shape[2], : pred.shape[3]].float())
    for loss in test_losses:
        print(loss.name, [round(float(v), 6) for v in loss(pred, gt).reshape(-1)], flush=True)


if __name__ == "__main__":
    main()
