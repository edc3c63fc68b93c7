#!/usr/bin/env python3
"""Counterpart of the reference driver examples/intermediate_downscaling.py for the MI355X HIP path.

Same CLI (`python intermediate_downscaling.py <config.yaml>`), same YAML schema (trainer / parallelism / tiling /
model / data, reference configs/interm_*.yaml), same environment contract (SLURM_NTASKS / SLURM_PROCID /
SLURM_LOCALID / HOSTNAME, with a torchrun-style RANK / WORLD_SIZE / LOCAL_RANK fallback), same per-step prints
and the same checkpoint dictionary keys and paths (reference :775-795).  Differences by decision:
  * the data-parallel engine is climate_learn.HipDataParallel (the FSDP NO_SHARD + bf16 MixedPrecision
    equivalent); `fsdp > 1` selects its sharded-optimizer mode, `tensor_par > 1` the head-split tensor-parallel
    blocks (climate_learn.dist.tp) with the reference's rank layout (tensor-parallel ranks adjacent) and per-rank
    checkpoint files `<ckpt>_rank_<r>`; seq_par must be 1 as in the reference;
  * the bf16 branch uses HipGradScaler(init_scale=8192, growth_interval=100, min_scale=128) -- the behaviour
    the reference intends at :493-497 (where it raises NameError);
  * data comes from the synthetic IterDataModule (the npz data plane is SURVEY 8f-1).
"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn as nn
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "orbit-2_amd"))

import climate_learn as cl  # noqa: E402
from climate_learn.data.processing.era5_constants import CONSTANTS  # noqa: E402,F401
from climate_learn.dist.profile import *  # noqa: E402,F401,F403
from climate_learn.models.hub.components.pos_embed import interpolate_pos_embed  # noqa: E402
from climate_learn.models.hub.components.vit_blocks import Block  # noqa: E402
from climate_learn.trainer import training_step, validation_step  # noqa: E402
from climate_learn.utils.fused_attn import FusedAttn  # noqa: E402


def seed_everything(seed, rank=0):
    """Python's and numpy's global streams get the SAME seed on every rank (the reference calls random.seed(0)
    everywhere); the data plane does not depend on them any more (private streams, data/iterdataset.py), torch's host
    stream is only used for weight init.  Only the kernel seed stream (dropout / DropPath masks) is keyed by the
    data-parallel rank."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    cl.manual_seed(seed, rank)


def init_par_groups(world_rank, data_par_size, tensor_par_size, seq_par_size, fsdp_size, simple_ddp_size, world_size,
                    num_heads=None):
    """Reference :161-262.  Rank layout: tensor-parallel ranks are adjacent (fastest varying), data-parallel ranks
    stride by tensor_par_size.  Returns (data_par_group, tensor_par_group); every rank creates every group, in the
    same order.  With fsdp > 1 AND simple_ddp > 1 (HYBRID_SHARD) the data-parallel ranks of a tensor-parallel column are cut
    as the reference does (:213-241): consecutive runs of fsdp ranks shard a model replica, ranks fsdp apart replicate a
    shard; the two groups are returned as `init_par_groups.fsdp_group` / `.simple_ddp_group` (None otherwise)."""
    assert seq_par_size == 1, "Sequence parallelism not implemented"
    assert data_par_size * seq_par_size * tensor_par_size == world_size, \
        "DATA_PAR_SIZE * SEQ_PAR_SIZE * TENSOR_PAR_SIZE must equal to world_size"
    if num_heads is not None:
        assert num_heads % tensor_par_size == 0, "model heads % tensor parallel size must be 0"
    init_par_groups.fsdp_group = init_par_groups.simple_ddp_group = None
    if world_size == 1:
        return None, None
    tensor_par_group = data_par_group = None
    if tensor_par_size > 1:
        for i in range(data_par_size):
            ranks = list(range(i * tensor_par_size, (i + 1) * tensor_par_size))
            group = dist.new_group(ranks)
            if world_rank in ranks:
                tensor_par_group = group
    for i in range(tensor_par_size):
        ranks = [i + j * tensor_par_size for j in range(data_par_size)]
        group = dist.new_group(ranks)
        if world_rank in ranks:
            data_par_group = group
        if fsdp_size > 1 and simple_ddp_size > 1:
            for k in range(simple_ddp_size):
                fr = ranks[k * fsdp_size:(k + 1) * fsdp_size]
                g = dist.new_group(fr)
                if world_rank in fr:
                    init_par_groups.fsdp_group = g
            for k in range(fsdp_size):
                dr = ranks[k::fsdp_size]
                g = dist.new_group(dr)
                if world_rank in dr:
                    init_par_groups.simple_ddp_group = g
    return data_par_group, tensor_par_group


def load_checkpoint(model, path, rank):
    if rank == 0:
        print("model resume from checkpoint", path, flush=True)
    return cl.utils.load_checkpoint(model, path)


def main():
    cfg_path = sys.argv[1]
    conf = yaml.load(open(cfg_path), Loader=yaml.FullLoader)
    world_size = int(os.environ.get("SLURM_NTASKS", os.environ.get("WORLD_SIZE", "1")))
    world_rank = int(os.environ.get("SLURM_PROCID", os.environ.get("RANK", "0")))
    local_rank = int(os.environ.get("SLURM_LOCALID", os.environ.get("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world_size > 1:
        # RCCL, one rank per GPU.  ORBIT2_DIST_BACKEND=gloo exists for rehearsing the multi-rank control flow (groups,
        # per-rank checkpoints, tensor-parallel collectives staged through the host) with several ranks on ONE card.
        backend = os.environ.get("ORBIT2_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=world_rank, world_size=world_size, device_id=device)
        else:
            dist.init_process_group(backend, rank=world_rank, world_size=world_size)

    tr, par, mc, dc = conf["trainer"], conf["parallelism"], conf["model"], conf["data"]
    max_epochs, batch_size = tr["max_epochs"], tr["batch_size"]
    data_type, train_loss_str = tr.get("data_type", "bfloat16"), tr["train_loss"]
    if data_type not in ("bfloat16", "float32"):
        raise RuntimeError("Data type not supported")
    tiling = conf.get("tiling", {}) or {}
    div, overlap = (tiling.get("div", 1), tiling.get("overlap", 0)) if tiling.get("do_tiling", False) else (1, 0)
    fsdp_size, ddp_size = par.get("fsdp", 1), par.get("simple_ddp", 1)
    tp, sp = par.get("tensor_par", 1), par.get("seq_par", 1)
    dp_size = fsdp_size * ddp_size
    dp_group, tp_group = init_par_groups(world_rank, dp_size, tp, sp, fsdp_size, ddp_size, world_size, mc["num_heads"])
    dp_rank, tp_rank = world_rank // tp, world_rank % tp

    # a tensor-parallel rank that overflows in ITS weight shard must make every rank skip the step: sync_world
    scaler = cl.HipGradScaler(init_scale=8192.0, growth_interval=100, min_scale=128.0,
                              sync_world=tp > 1) if data_type == "bfloat16" else None
    model = eng = optimizer = scheduler = None
    epoch_start = 0
    seed_everything(0, dp_rank)       # the ranks of one tensor-parallel group share data and dropout seeds
    for data_key in dc["low_res_dir"]:
        in_vars, out_vars = dc["dict_in_variables"][data_key], dc["dict_out_variables"][data_key]
        syn = (dc.get("synthetic") or {}).get(data_key, {})
        dm = cl.data.IterDataModule(
            "downscaling", dc["low_res_dir"][data_key], dc["high_res_dir"][data_key], in_vars, out_vars=out_vars,
            data_par_size=dp_size, data_par_group=dp_group, subsample=1, batch_size=batch_size,
            buffer_size=tr.get("buffer_size", 0), num_workers=tr.get("num_workers", 0), div=div, overlap=overlap,
            lowres_hw=tuple(syn.get("lowres_hw", (32, 64))), highres_hw=tuple(syn["highres_hw"]) if "highres_hw" in syn else None,
            steps_per_epoch=syn.get("steps_per_epoch", 4))
        dm.setup()
        if model is None:
            with torch.device(device):
                out = cl.load_downscaling_module(
                    device, data_module=dm, architecture=mc["preset"], train_loss=train_loss_str,
                    model_kwargs={"default_vars": dc["default_vars"], "superres_mag": mc["superres_mag"],
                                  "cnn_ratio": mc["cnn_ratio"], "patch_size": mc["patch_size"],
                                  "embed_dim": mc["embed_dim"], "depth": mc["depth"],
                                  "decoder_depth": mc["decoder_depth"], "num_heads": mc["num_heads"],
                                  "mlp_ratio": mc["mlp_ratio"], "drop_path": mc["drop_path"],
                                  "drop_rate": mc["drop_rate"], "tensor_par_size": tp, "tensor_par_group": tp_group,
                                  "FusedAttn_option": FusedAttn.CK if data_type == "bfloat16" else FusedAttn.DEFAULT})
            model, train_loss = out[0], out[1]
            val_losses, val_transforms = out[2], out[5]
            ck = None
            suffix = "_rank_" + str(tp_rank) if tp > 1 else ""      # per tensor-parallel rank files (reference :52,:72)
            if tr.get("checkpoint") and os.path.exists(str(tr["checkpoint"]) + suffix):
                ck = load_checkpoint(model, str(tr["checkpoint"]) + suffix, world_rank)
            elif tr.get("pretrain"):       # shape-tolerant partial load (reference :70-80, :116-153)
                if world_rank == 0:
                    print("load pretrained model", tr["pretrain"], flush=True)
                cl.utils.load_pretrained_weights(model, str(tr["pretrain"]) + suffix, verbose=world_rank == 0)
            elif tp > 1:                   # from scratch: replicated weights come from the group's first rank (:83-112)
                cl.dist.tp.sync_replicated(model, tp_group)
            # parallelism.fsdp > 1 asks the reference for sharded FSDP (:609-617): FULL_SHARD over the data-parallel ranks, or
            # HYBRID_SHARD (shards of `fsdp` ranks, replicated `simple_ddp` times) -> the parameter-sharding engine.
            # `parallelism.shard_strategy: grad_op` keeps the parameters replicated and shards gradients + optimizer state only
            # (the SHARD_GRAD_OP-like mode of the NO_SHARD engine).  Both combine with tensor parallelism: the reference's
            # 2-D layout (configs/interm_1b.yaml:14-24, fsdp x simple_ddp x tensor_par) shards every tensor-parallel column
            # over its own data-parallel ranks.
            shard = fsdp_size > 1 and dp_size > 1
            strategy = par.get("shard_strategy", "full")
            if strategy not in ("full", "grad_op"):
                raise ValueError("parallelism.shard_strategy must be 'full' or 'grad_op', got %r" % (strategy,))
            full = shard and strategy == "full"
            if full:
                hybrid = ddp_size > 1
                print("enter hybrid FSDP," if hybrid else "enter fully sharded FSDP,", flush=True)
                eng = cl.HipFullyShardedDataParallel(
                    model, process_group=init_par_groups.fsdp_group if hybrid else dp_group, unit_types=(Block, nn.Sequential),
                    sync_module_states=True, replicate_group=init_par_groups.simple_ddp_group if hybrid else None,
                    tp_group=tp_group)
                if world_rank == 0:
                    print("per-rank parameter bytes:", eng.param_bytes_per_rank(), flush=True)
            else:
                print("enter sharded optimizer (SHARD_GRAD_OP-like)," if shard else "enter NO SHARD only,", flush=True)
                eng = cl.HipDataParallel(model, process_group=dp_group, unit_types=(Block, nn.Sequential),
                                         sync_module_states=True, shard_optimizer=shard, replica_group=tp_group)
            for blk in model.blocks:
                blk.recompute = bool(tr.get("activation_checkpointing", False))
            optimizer = cl.load_optimizer(eng, "adamw", {"lr": float(mc["lr"]), "weight_decay": float(mc["weight_decay"]),
                                                         "betas": (mc["beta_1"], mc["beta_2"])})
            scheduler = cl.load_lr_scheduler("linear-warmup-cosine-annealing", optimizer,
                                             {"warmup_epochs": mc["warmup_epochs"], "max_epochs": max_epochs,
                                              "warmup_start_lr": float(mc["warmup_start_lr"]),
                                              "eta_min": float(mc["eta_min"])})
            if ck is not None:
                optimizer.load_state_dict(ck["optimizer_state_dict"])
                scheduler.load_state_dict(ck["scheduler_state_dict"])
                epoch_start = ck["epoch"] + 1
        in_shape, _ = dm.get_data_dims()
        eng.data_config(dc["spatial_resolution"][data_key], tuple(in_shape[2:]), len(in_vars), len(out_vars))
        var_weights = dc.get("var_weights", {})
        # hipGraph replay of zero_grad + forward + loss + backward (climate_learn/graphs.py) for the launch-bound
        # configurations: `trainer.hipgraph: auto` (default) turns it on when a step has few tokens (<= 16384: the 32x64-grid
        # presets; interm_1b-class steps are GPU-bound and stay eager), true / false force it.  Batches of another shape than
        # the captured one (the short last batch of an epoch) run eagerly.
        hg = tr.get("hipgraph", "auto")
        tokens = batch_size * (in_shape[2] // mc["patch_size"]) * (in_shape[3] // mc["patch_size"])
        capturable = world_size == 1 or dist.get_backend() == "nccl"      # gloo rehearsals stage through the host: no capture
        capturable = capturable and getattr(train_loss, "graph_capturable", True)
        # the parameter-sharding engine is captured in its single-stream form (round 6: fsdp_engine.single_stream) --
        # on request only (`hipgraph: true`): `auto` keeps it eager, its models are not launch-bound
        sharded = getattr(eng, "shard_params", False)
        if hg is True and (tp > 1 or not capturable):
            raise ValueError("trainer.hipgraph: true is not available here (tensor parallelism or a gloo rehearsal): use 'auto' or "
                             "false")     # never a silent fallback
        use_graph = (hg is True or (hg == "auto" and tokens <= 16384 and capturable and not sharded)) and tp == 1
        gstep, gshape = None, None
        for epoch in range(epoch_start, max_epochs):
            eng.train()
            epoch_loss = torch.zeros((), dtype=torch.float32, device=device)
            if world_rank == 0:
                print("epoch ", epoch, flush=True)
            for batch_idx, batch in enumerate(dm.train_dataloader()):
                if world_rank == 0:
                    torch.cuda.synchronize(device)
                    tic1 = time.perf_counter()
                shape = (tuple(batch[0].shape), tuple(batch[1].shape))
                if use_graph and (gstep is None or shape == gshape):
                    if gstep is None:
                        gstep, gshape = cl.GraphedTrainStep(eng, train_loss, batch, var_weights, scaler=scaler), shape
                    loss = gstep(batch)
                    epoch_loss += loss
                    if world_rank == 0:
                        print("epoch: ", epoch, "batch_idx", batch_idx, "world_rank", world_rank, " loss ", float(loss), flush=True)
                    if scaler is None:
                        optimizer.step()
                    else:
                        scaler.step(optimizer)
                        scaler.update()
                else:
                    loss = training_step(batch, batch_idx, eng, device, var_weights, train_loss)
                    epoch_loss += loss.detach()
                    if world_rank == 0:
                        print("epoch: ", epoch, "batch_idx", batch_idx, "world_rank", world_rank, " loss ", float(loss), flush=True)
                    optimizer.zero_grad()
                    if scaler is None:
                        loss.backward()
                        optimizer.step()
                    else:
                        scaler.scale(loss).backward()
                        scaler.step(optimizer)
                        scaler.update()
                if world_rank == 0:
                    torch.cuda.synchronize(device)
                    print("rank", world_rank, "batch_idx", batch_idx, "get_lr ", scheduler.get_last_lr(),
                          "after optimizer step torch.cuda.memory_reserved: %fGB" % (torch.cuda.memory_reserved(device) / 2 ** 30),
                          flush=True)
                    print(f"my rank {world_rank}. tic4-tic1 in {(time.perf_counter() - tic1):0.4f} seconds\n", flush=True)
            scheduler.step()
            if world_rank == 0:
                print("epoch: ", epoch, " epoch_loss ", float(epoch_loss), flush=True)
            if getattr(eng, "shard_params", False):      # assembled unit by unit onto the host: no full fp32 model + moments on the GPU
                model_states, optimizer_states = eng.state_dict(offload_to_cpu=True), optimizer.state_dict(offload_to_cpu=True)
            else:
                model_states, optimizer_states = eng.state_dict(), optimizer.state_dict()     # collective when the optimizer is sharded
            if world_rank < tp:            # one file per tensor-parallel rank of the first group (reference :778-790)
                os.makedirs("checkpoints/climate", exist_ok=True)
                torch.save({"epoch": epoch, "model_state_dict": model_states,
                            "optimizer_state_dict": optimizer_states,
                            "scheduler_state_dict": scheduler.state_dict()},
                           "checkpoints/climate/interm_epoch_" + str(epoch) + ".ckpt" + suffix)
            del model_states, optimizer_states
            # validation pass: rmse / pearson / mean_bias on denormalised fields + mse, eval mode.  The reference has the
            # block but switches it off (`if False:`, :801-822); `trainer.validate: true` switches it on here.
            if tr.get("validate", False):
                with torch.no_grad():
                    eng.eval()
                    if world_rank == 0:
                        print("val epoch ", epoch, flush=True)
                    for batch_idx, batch in enumerate(dm.val_dataloader()):
                        losses = validation_step(batch, batch_idx, eng, device, val_losses, val_transforms)
                        if world_rank == 0:
                            print("val epoch: ", epoch, "batch_idx", batch_idx, "world_rank", world_rank, " losses ",
                                  {k: round(float(v), 6) for k, v in losses.items()}, flush=True)
            if world_size > 1:
                dist.barrier()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
