"""CPU tests of the host-side mirror of the reference interface (no kernels run)."""
import os

import numpy as np
import pytest
import torch

CONST = ["land_sea_mask", "orography", "lattitude", "landcover"]


def test_state_dict_keys_and_shapes_match_reference_checkpoint_format(golden_dir):
    from climate_learn.models.hub import Res_Slim_ViT, MODEL_REGISTRY
    z = np.load(os.path.join(golden_dir, "model_v7c3_hd64.npz"))
    ref = {k[2:]: z[k].shape for k in z.files if k.startswith("p.")}
    dv = CONST + ["2m_temperature", "total_precipitation_24hr", "2m_temperature_min", "2m_temperature_max"]
    m = Res_Slim_ViT(dv, (16, 32), 7, 3, 1, patch_size=2, embed_dim=128, depth=1, decoder_depth=2, num_heads=2)
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == {k: tuple(v) for k, v in ref.items()}
    assert MODEL_REGISTRY["res_slimvit"] is Res_Slim_ViT
    # pos_embed init = the reference's sincos table
    assert np.allclose(m.pos_embed.detach().numpy(), z["p.pos_embed"], atol=0.2)  # fixture adds N(0, .02) noise


def test_param_count_closed_form():
    from climate_learn.models.hub import Res_Slim_ViT
    # interm_8m with V=23, C=3, init grid 32x64 -> 5 485 443 parameters
    dv = ["v%d" % i for i in range(23)]
    m = Res_Slim_ViT(dv, (32, 64), 23, 3, 1, patch_size=2, embed_dim=256, depth=6, decoder_depth=4, num_heads=4)
    assert sum(p.numel() for p in m.parameters()) == 5485443


def test_model_refuses_cpu_input():
    from climate_learn.models.hub import Res_Slim_ViT
    m = Res_Slim_ViT(CONST + ["total_precipitation_24hr"], (16, 32), 5, 1, 1, patch_size=2, embed_dim=128, depth=1,
                     decoder_depth=1, num_heads=2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 5, 16, 32), CONST + ["total_precipitation_24hr"], ["total_precipitation_24hr"])
    with pytest.raises(ValueError):
        m.find_var_index(["a", "b"], ["a"])


def test_loader_api_errors_and_tuple():
    import climate_learn as cl
    dm = cl.data.IterDataModule("downscaling", "lo", "hi", CONST + ["total_precipitation_24hr"],
                                ["total_precipitation_24hr"], batch_size=2, lowres_hw=(16, 32))
    with pytest.raises(RuntimeError, match="Data module has not been set up yet."):
        cl.load_downscaling_module("cpu", data_module=dm, architecture="res_slimvit")
    dm.setup()
    out = cl.load_downscaling_module(
        "cpu", data_module=dm, architecture="res_slimvit", train_loss="bayesian_tv",
        model_kwargs={"default_vars": CONST + ["total_precipitation_24hr"], "embed_dim": 128, "depth": 1,
                      "decoder_depth": 1, "num_heads": 2, "FusedAttn_option": cl.FusedAttn.CK})
    assert len(out) == 7
    model, train_loss = out[0], out[1]
    assert train_loss.name == "bayesian_tv" and train_loss.aggregate_only
    assert model.img_size == (16, 32) and model.out_channels == 1
    assert model.blocks[0].attn.attn_p() == 0.1      # CK semantics: P-dropout even outside training
    model.eval()
    assert model.blocks[0].attn.attn_p() == 0.1
    with pytest.raises(NotImplementedError):
        cl.load_loss("cpu", model, "no_such_loss", True, None)
    with pytest.raises(NotImplementedError):
        cl.load_optimizer(model, "sgd")
    with pytest.raises(NotImplementedError):
        cl.load_lr_scheduler("foo", object())
    x, y, iv, ov = dm.train_dataloader()[0]
    assert x.shape == (2, 5, 16, 32) and y.shape == (2, 1, 64, 128) and float(y.min()) >= 0.0


def test_lr_scheduler_matches_reference_golden(golden_dir):
    import climate_learn as cl
    z = np.load(os.path.join(golden_dir, "lr_schedule.npz"))
    p = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.SGD(p, lr=5e-4)
    sc = cl.load_lr_scheduler("linear-warmup-cosine-annealing", opt,
                              {"warmup_epochs": 2, "max_epochs": 100, "warmup_start_lr": 1e-7, "eta_min": 1e-8})
    lrs = []
    for _ in range(100):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sc.step()
    assert np.allclose(lrs, z["lr_w2_m100"], rtol=1e-9, atol=1e-15)


def test_hash_replica_known_answers():
    # pins tests/hashmask.py (and through the GPU tests, csrc/common.h:o2_hash64) to fixed values
    from tests.hashmask import o2_hash64, keep_mask
    h = o2_hash64(0x1234, np.array([0, 1, 2**33 + 5], dtype=np.uint64))
    assert h.dtype == np.uint64 and len(set(h.tolist())) == 3
    m, sc = keep_mask(7, 1 << 16, 0.1)
    assert abs(m.mean() - (1 - 26 / 256)) < 5e-3 and abs(sc - 256 / 230) < 1e-12


def test_attention_dropout_mask_statistics():
    """the factored attention mask hash (one multiply after R(row) ^ K(key group)) must look like independent
    Bernoulli draws: keep rate, neighbour correlations along both axes, the parity of 2x2 row/key-group rectangles
    (the structure a plain xor of two hashes would leave) and the per-row / per-key keep-count variances"""
    from tests.hashmask import attn_keep_mask
    m, sc = attn_keep_mask(0x1234567ABCDEF, 2, 1024, 0.1)
    m = m.reshape(-1, m.shape[-1]).astype(np.float64)               # [rows, keys]
    keep = 1 - 26 / 256
    assert abs(m.mean() - keep) < 2e-3 and abs(sc - 256 / 230) < 1e-12
    c = lambda a, b: float(((a - keep) * (b - keep)).mean() / (keep * (1 - keep)))
    tol = 5 / np.sqrt(m.size)                                        # ~5 sigma of an independent field
    assert abs(c(m[:-1], m[1:])) < tol and abs(c(m[:, :-1], m[:, 1:])) < tol and abs(c(m[:, :-4], m[:, 4:])) < tol
    par = (m[:-1, :-4] + m[1:, 4:] + m[:-1, 4:] + m[1:, :-4]) % 2
    q = 1 - keep
    assert abs(par.mean() - (1 - (1 - 2 * q) ** 4) / 2) < 3e-3
    n = m.shape[1]
    assert 0.85 < m.sum(1).var() / (n * keep * (1 - keep)) < 1.15
    assert 0.85 < m.sum(0).var() / (m.shape[0] * keep * (1 - keep)) < 1.15


def test_checkpoint_pos_embed_interpolation_matches_oracle():
    """interpolate_pos_embed (reference components/pos_embed.py:75-100) on a checkpoint dict, vs the oracle."""
    from types import SimpleNamespace
    from climate_learn.models.hub.components.pos_embed import interpolate_pos_embed, interpolate_pos_embed_on_the_fly
    from oracle import orbit2_oracle as O
    g = torch.Generator().manual_seed(5)
    pe = torch.randn(1, 4 * 8, 64, generator=g)
    ck = {"pos_embed": pe.clone()}
    interpolate_pos_embed(SimpleNamespace(patch_size=2), ck, new_size=(16, 32))
    ref = O.pos_embed_for_grid(pe, 2, (16, 32))
    assert ck["pos_embed"].shape == (1, 8 * 16, 64) and torch.allclose(ck["pos_embed"], ref, atol=1e-6)
    assert torch.allclose(interpolate_pos_embed_on_the_fly(pe, 2, (16, 32)), ref, atol=1e-6)
    same = {"pos_embed": pe.clone()}
    interpolate_pos_embed(SimpleNamespace(patch_size=2), same, new_size=(8, 16))
    assert torch.equal(same["pos_embed"], pe)


def test_shape_tolerant_pretrain_load(tmp_path):
    """reference examples/intermediate_downscaling.py:116-153: unknown keys dropped, mismatched shapes dropped except
    pos_embed (bicubic resample to the model grid), the rest loaded non-strictly"""
    from climate_learn.models.hub import Res_Slim_ViT
    from climate_learn.utils import load_pretrained_weights, load_checkpoint
    from oracle import orbit2_oracle as O
    dv = CONST + ["total_precipitation_24hr"]
    torch.manual_seed(0)
    pre = Res_Slim_ViT(dv, (8, 16), 5, 1, 1, patch_size=2, embed_dim=128, depth=1, decoder_depth=1, num_heads=2)
    sd = {k: v.clone() + 0.01 for k, v in pre.state_dict().items()}
    sd["extra.not_in_model"] = torch.zeros(3)
    path = os.path.join(tmp_path, "pre.ckpt")
    torch.save({"epoch": 3, "model_state_dict": sd}, path)
    # fine-tune model: another grid (pos_embed 32 -> 128 tokens) and 3 output channels (head / conv shapes differ)
    dv3 = dv + ["2m_temperature_min", "2m_temperature_max"]
    torch.manual_seed(1)
    ft = Res_Slim_ViT(dv3, (16, 32), 7, 3, 1, patch_size=2, embed_dim=128, depth=1, decoder_depth=1, num_heads=2)
    before = {k: v.clone() for k, v in ft.state_dict().items()}
    loaded, no_key, bad = load_pretrained_weights(ft, path)
    after = ft.state_dict()
    assert no_key == ["extra.not_in_model"]
    assert "pos_embed" in loaded and "pos_embed" not in bad
    assert torch.allclose(after["pos_embed"], O.pos_embed_for_grid(sd["pos_embed"], 2, (16, 32)), atol=1e-6)
    assert "blocks.0.attn.qkv.weight" in loaded and torch.equal(after["blocks.0.attn.qkv.weight"], sd["blocks.0.attn.qkv.weight"])
    assert len(bad) > 0 and all(sd[k].shape != before[k].shape for k in bad)
    for k in bad:                                              # untouched: still the fine-tune model's own init
        assert torch.equal(after[k], before[k])
    with pytest.raises(SystemExit):
        load_pretrained_weights(ft, os.path.join(tmp_path, "missing.ckpt"))
    # strict resume of the same architecture goes through load_checkpoint (pos_embed resample included)
    torch.save({"epoch": 3, "model_state_dict": {k: v for k, v in sd.items() if k != "extra.not_in_model"}}, path)
    ck = load_checkpoint(pre, path)
    assert ck["epoch"] == 3 and torch.equal(pre.state_dict()["head.0.weight"], sd["head.0.weight"])


def test_weight_gradient_group_balance_plan():
    """host logic of the balanced grouped weight-gradient launch (_ops._dw_balance_plan): the tiles beyond the last whole round
    of 256 leave the full-length set as whole problems / whole tile rows and are split over the tokens so that both parts fill
    whole rounds; no exact plan -> None (the launch stays as it was)"""
    from climate_learn._ops import _dw_balance_plan
    # interm_1b Block: qkv 36 x 12, proj 12 x 12, fc1 48 x 12, fc2 12 x 48 = 1728 tiles = 6.75 rounds
    shapes = [(36, 12), (12, 12), (48, 12), (12, 48)]
    plan = _dw_balance_plan(shapes, 4)
    assert plan == [(1, 0, 12), (0, 32, 4)]                       # all of proj (144) + the last 4 tile rows of qkv (48) = 192
    cut = sum(r * shapes[i][1] for i, _, r in plan)
    total = sum(a * b for a, b in shapes)
    assert cut == total % 256 and (total - cut) % 256 == 0 and (cut * 4) % 256 == 0
    assert _dw_balance_plan(shapes, 2) is None                    # 192 x 2 is not a whole number of rounds
    assert _dw_balance_plan([(4, 4), (4, 4)], 4) is None          # fewer tiles than one round
    assert _dw_balance_plan([(16, 16)], 4) is None                # already whole rounds
    assert _dw_balance_plan([(36, 12), (12, 12), (48, 12), (12, 48)], 0) is None


def test_round_major_tile_ids_are_a_bijection():
    """mirror of csrc/gemm.hip:xcd_round_tile_id (block b runs on XCD b & 7; the 256 blocks of a round take 256 consecutive ids,
    each XCD a consecutive run of them) and of gemm256w_tile's id -> (tm, tn) walk incl. the column-major form of wide problems:
    every tile exactly once for any grid size, a round's ids contiguous, an XCD's ids of a round contiguous"""
    def rid(b, nwg):
        base = b & ~255
        cnt = min(256, nwg - base)
        x, s = b & 7, (b & 255) >> 3
        q8, r8 = cnt >> 3, cnt & 7
        return base + (x * (q8 + 1) if x < r8 else r8 * (q8 + 1) + (x - r8) * q8) + s

    for nwg in (1, 7, 8, 9, 255, 256, 257, 300, 511, 512, 1000, 1728, 6144, 6145):
        ids = [rid(b, nwg) for b in range(nwg)]
        assert sorted(ids) == list(range(nwg)), nwg
        for r0 in range(0, nwg, 256):
            rnd = ids[r0:r0 + 256]
            assert min(rnd) == r0 and max(rnd) == min(nwg, r0 + 256) - 1
            for x in range(8):
                mine = sorted(i for b, i in zip(range(r0, r0 + len(rnd)), rnd) if b & 7 == x)
                assert mine == list(range(mine[0], mine[0] + len(mine))) if mine else True

    def walk(i, tiles_m, tiles_n, group=4):
        tw = tiles_n > tiles_m
        t_major, t_minor = (tiles_n, tiles_m) if tw else (tiles_m, tiles_n)
        per_group = group * t_minor
        first = (i // per_group) * group
        gsz = min(group, t_major - first)
        ta, tb = first + (i % per_group) % gsz, (i % per_group) // gsz
        return (tb, ta) if tw else (ta, tb)

    for tm, tn in ((36, 12), (12, 12), (48, 12), (12, 48), (5, 3), (3, 5), (1, 9), (512, 12), (7, 7)):
        seen = {walk(i, tm, tn) for i in range(tm * tn)}
        assert seen == {(a, b) for a in range(tm) for b in range(tn)}, (tm, tn)
