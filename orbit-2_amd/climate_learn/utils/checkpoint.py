"""Checkpoint / pretrain loading of the reference's `.ckpt` dictionaries
({'epoch', 'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict'}; reference
examples/intermediate_downscaling.py:45-153, 775-795)."""
import os
from typing import Dict, List, Tuple

import torch

from ..models.hub.components.pos_embed import interpolate_pos_embed


def load_pretrained_weights(model, pretrain_path: str, verbose: bool = False) -> Tuple[List[str], List[str], List[str]]:
    """Shape-tolerant partial load (reference `_load_pretrained_weights`, :116-153): keys the model does not have are
    dropped, keys whose shape differs are dropped -- except `pos_embed`, which is resampled bicubically to the
    model's grid -- and the rest is loaded with strict=False.
    Returns (loaded, dropped_missing_in_model, dropped_shape_mismatch)."""
    if not os.path.exists(pretrain_path):
        raise SystemExit("pretrain path does not exist")           # the reference exits with this message (:80)
    ck = torch.load(pretrain_path, map_location="cpu")
    src: Dict[str, torch.Tensor] = dict(ck["model_state_dict"])
    del ck
    own = model.state_dict()
    no_key, bad_shape = [], []
    for k in list(src.keys()):
        if k not in own:
            no_key.append(k)
            del src[k]
        elif src[k].shape != own[k].shape:
            if k == "pos_embed":
                interpolate_pos_embed(model, src, new_size=model.img_size)
                if src[k].shape != own[k].shape:                   # grids of another aspect: cannot be resampled here
                    bad_shape.append(k)
                    del src[k]
            else:
                bad_shape.append(k)
                del src[k]
    msg = model.load_state_dict(src, strict=False)
    if verbose:
        for k in no_key:
            print(f"Removing key {k} from pretrained checkpoint: no exist")
        for k in bad_shape:
            print(f"Removing key {k} from pretrained checkpoint: no matching shape")
        print(msg)
    return sorted(src.keys()), no_key, bad_shape


def load_checkpoint(model, path: str):
    """Resume (reference :50-68): the full model_state_dict must match; returns the checkpoint dict so the caller can
    restore optimizer / scheduler / epoch."""
    if not os.path.exists(path):
        raise SystemExit("checkpoint path does not exist")
    ck = torch.load(path, map_location="cpu")
    sd = ck["model_state_dict"]
    interpolate_pos_embed(model, sd, new_size=model.img_size)
    model.load_state_dict(sd)
    return ck
