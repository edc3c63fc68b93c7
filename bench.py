#!/usr/bin/env python3
"""Headline benchmark: climate-grid samples/s (fwd + loss + bwd + DP all-reduce + AdamW) of the HIP
Res_Slim_ViT training step.  Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
launched as one rank per GPU by torch.distributed.run.  Rank 0 prints ONE JSON line.

Workload at N=1 (and per GPU at N>1, weak scaling): BASELINE.json configs[2] -- interm_1b (D3072/depth 8/
24 heads/decoder 4), ERA5 1.40625deg -> 0.25deg synthetic grids: x [B,23,128,256] -> pred [B,3,512,1024],
target [B,3,721,1440] consumed through its top-left 512x1024 crop, bf16 compute / fp32 master, train mode with
the YAML's dropout 0.1 / drop-path 0.1, loss bayesian_tv, dynamic loss scaling, fused AdamW.
"""
import argparse
import json
import os
import sys
import time

T_PROCESS_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "orbit-2_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

CONST = ["land_sea_mask", "orography", "lattitude", "landcover"]
ERA5_VARS = CONST + [
    "2m_temperature", "2m_temperature_max", "2m_temperature_min", "temperature_200", "temperature_500",
    "temperature_850", "10m_u_component_of_wind", "u_component_of_wind_200", "u_component_of_wind_500",
    "u_component_of_wind_850", "10m_v_component_of_wind", "v_component_of_wind_200", "v_component_of_wind_500",
    "v_component_of_wind_850", "specific_humidity_200", "specific_humidity_500", "specific_humidity_850",
    "total_precipitation_24hr", "volumetric_soil_water_layer_1"]
OUT_VARS = ["total_precipitation_24hr", "2m_temperature_min", "2m_temperature_max"]
VAR_WEIGHTS = {"2m_temperature": 10, "10m_u_component_of_wind": 1, "10m_v_component_of_wind": 1,
               "total_precipitation_24hr": 1, "2m_temperature_min": 10, "2m_temperature_max": 10}
MODELS = {  # configs/interm_*.yaml of the reference
    "interm_8m": dict(embed_dim=256, depth=6, num_heads=4),
    "interm_117m": dict(embed_dim=1024, depth=8, num_heads=16),
    "interm_1b": dict(embed_dim=3072, depth=8, num_heads=24),
    "interm_10b": dict(embed_dim=8192, depth=11, num_heads=32),
}
PEAK_BF16 = 2.5e15   # dense MFMA peak, MI355X_MICROARCH.md
MALL_JSON = "r06_mall_latency.json"  # Infinity-Cache share of that traffic (TCC_EA0_RDREQ_LEVEL pass of tools/mall_probe.py)
TRAFFIC_JSON = "r06_traffic.json"   # PMC passes (FETCH_SIZE / WRITE_SIZE) of the bench configuration, tools/profile_round.sh
METRIC = "climate-grid samples/sec/node (fwd+bwd), interm_1b ERA5 1.4°→0.25°, 1/2/4/8 GPUs"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="interm_1b")
    ap.add_argument("--grid", default="128x256")
    ap.add_argument("--batch", type=int, default=16,
                    help="per-GPU batch: 16 (1 / 2 / 4 / 8 / 16 / 32 all fit one GPU; 8 is 1 % slower, 32 = the reference "
                         "YAML's batch_size gives the same rate at twice the step time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "npz"],
                    help="npz: after the resident-batch measurement, the SAME step fed through the reference-format data plane "
                         "(a synthesized <root>/{train}/<year>_<shard>.npz tree at the run's grids -> climate_learn.data.IterDataModule "
                         ".train_dataloader() with worker processes and pinned batches -> non_blocking uploads); attached as `data_npz`")
    ap.add_argument("--data-workers", type=int, default=8)
    ap.add_argument("--other-configs-smoke", action="store_true",
                    help="tests: attach `other_configs` to ANY single-GPU run, with one tiny child run (interm_8m) instead of the three")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the child runs of the other BASELINE configs (interm_117m 32x64 batch 8, the Daymet-like interm_1b "
                         "step, interm_10b batch 2) that the default single-GPU run attaches as `other_configs`")
    ap.add_argument("--no-comm-stats", action="store_true",
                    help="do not attach CommStats (timing events + collective waits on the communication stream) to the engine: "
                         "A/B of what the accounting itself costs (tools/forced_collectives_ab.sh)")
    ap.add_argument("--no-fused-colsum", action="store_true",
                    help="fc1's bias gradient by a separate column-sum pass over the GELU input gradient instead of the sums its "
                         "producing GEMM leaves per tile row (A/B)")
    ap.add_argument("--eager-baseline", type=int, default=0, metavar="B",
                    help="also time the oracle (the plain-PyTorch restatement of the reference) ON THE GPU under bf16 autocast "
                         "with the framework's fused attention, per-GPU batch B: what the reference's eager PyTorch step costs "
                         "on this card (reported as `gpu_eager_baseline`; off by default)")
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--recompute", action="store_true", help="replay each Block in backward (activation ckpt)")
    ap.add_argument("--daymet", action="store_true",
                    help="BASELINE configs[4] / SURVEY 8d-5: 7 Daymet-like inputs, 3 outputs, hybrid perceptual loss "
                         "(use --grid 96x192); not the headline configuration")
    ap.add_argument("--mall-probe", action="store_true",
                    help="run the memory-side latency calibration streams (orbit2_probe_read) before the steps: for the "
                         "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum pass of tools/profile_round.sh")
    ap.add_argument("--daymet-loss", default="perceptual_lat_mse", choices=["perceptual_lat_mse", "perceptual"],
                    help="loss of the --daymet configuration: the hybrid perceptual + latitude-weighted MSE sum BASELINE "
                         "configs[4] names (default), or the reference's `perceptual` object alone")
    ap.add_argument("--graph", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                    help="replay zero_grad+forward+loss+backward (+bucket all-reduces) from one captured hipGraph; auto "
                         "(default): on when a step has <= 16384 tokens per GPU (launch-bound small configurations), off "
                         "for the GPU-bound headline configuration, whose kernels are then timed with HIP events inside "
                         "the timed region; with the graph on the roofline fields come from an eager instrumented pass")
    ap.add_argument("--tensor-par", type=int, default=1,
                    help="head-split tensor parallelism over adjacent ranks (DESIGN 5c); WORLD_SIZE = dp x tp, the "
                         "reported value counts dp x batch samples per step")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous of the N ranks only (gloo, no GPU call, no workload): checks the self-launch path")
    ap.add_argument("--fsdp", action="store_true",
                    help="parameter sharding over the data-parallel ranks (the reference's FSDP FULL_SHARD): per-unit "
                         "all-gather one unit ahead, reduce-scattered gradients, AdamW on 1/N chunks")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="reduce-scatter gradients, AdamW on 1/N of every unit, all-gather the bf16 copies")
    return ap.parse_args()


def gpu_eager_baseline(model_name, V, C, B, dev):
    """The oracle's math as eager PyTorch on the GPU: bf16 autocast, hipBLASLt GEMMs, SDPA flash attention with dropout
    0.1, torch.optim.AdamW(fused) on fp32 parameters -- the reference's own formulation (per-variable patch-embed
    tokens + kv GEMM over B*L*V rows, unfused epilogues) at the full 128x256 grid.  A baseline beside the
    measurement, never the measurement."""
    import torch.nn.functional as F
    from oracle import orbit2_oracle as O
    m = MODELS[model_name]
    cfg = O.Config(ERA5_VARS, (128, 256), C, m["embed_dim"], m["depth"], 4, m["num_heads"], spatial_resolution=156.0)
    sd = {k: v.to(dev).requires_grad_() for k, v in O.init_state_dict(cfg, V, seed=0, fast=True).items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=5e-4, betas=(0.9, 0.99), weight_decay=1e-5, fused=True)
    naive = O.mha_core

    def flash(q, k, v, scale, pmask=None):
        if q.shape[-2] != k.shape[-2]:                 # the 1-query variable aggregation stays on the plain form
            return naive(q, k, v, scale, pmask)
        return F.scaled_dot_product_attention(q, k, v, dropout_p=0.1, scale=scale)
    O.mha_core = flash
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, V, 128, 256, generator=g).to(dev)
    y = torch.randn(B, C, 721, 1440, generator=g).abs().to(dev)

    def step():
        with torch.device(dev), torch.autocast("cuda", dtype=torch.bfloat16):     # the oracle's constants follow the device
            loss = O.training_loss(sd, cfg, x, y, ERA5_VARS, OUT_VARS, "bayesian_tv", VAR_WEIGHTS)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    try:
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            last = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        return {"value": B / dt, "unit": "samples/s", "ms_per_step": 1e3 * dt, "per_gpu_batch": B, "kind": "port",
                "sample": "oracle (plain PyTorch restatement of the reference) on the same GPU: bf16 autocast, library "
                          "GEMMs, SDPA attention with dropout 0.1, fused torch AdamW; no other dropout / DropPath; 3 timed steps",
                "final_loss": float(last.detach())}
    finally:
        O.mha_core = naive


def _host_threads():
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 16))          # the GPU box grants a 16-core CPU share per GPU


def _oracle_steps(model_name, in_vars, out_vars, B, grid, loss, warm, timed, tag):
    """fwd + loss + bwd + AdamW of the oracle (plain PyTorch fp32 restatement of the reference, kind 'port') on the host:
    `warm` untimed steps, then `timed` steps timed one by one; returns the list of step times"""
    from oracle import orbit2_oracle as O
    m = MODELS[model_name]
    V, C = len(in_vars), len(out_vars)
    cfg = O.Config(in_vars, grid, C, m["embed_dim"], m["depth"], 4, m["num_heads"], spatial_resolution=156.0)
    g = torch.Generator().manual_seed(0)
    sd = {k: v.requires_grad_() for k, v in O.init_state_dict(cfg, V, seed=0, fast=True).items()}
    mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in sd.items()}
    x = torch.randn(B, V, *grid, generator=g)
    y = torch.randn(B, C, 4 * grid[0], 4 * grid[1], generator=g).abs()
    names = list(sd)

    def step(i):
        loss_v = O.training_loss(sd, cfg, x, y, in_vars, out_vars, loss, VAR_WEIGHTS)
        grads = torch.autograd.grad(loss_v, [sd[k] for k in names], allow_unused=True)
        with torch.no_grad():
            for k, gr in zip(names, grads):
                if gr is not None:
                    O.adamw_step(sd[k], gr, mom[k][0], mom[k][1], i, 5e-4, 0.9, 0.99, 1e-8, 1e-5)

    ts = []
    for i in range(warm + timed):
        t0 = time.perf_counter()
        step(i + 1)
        dt = time.perf_counter() - t0
        if i >= warm:
            ts.append(dt)
        print("[bench] cpu_baseline %s: step %d/%d %.2f s" % (tag, i + 1, warm + timed, dt), file=sys.stderr, flush=True)
    return ts


def cpu_baseline(model_name, V, C):
    """The oracle timed on the GPU box's host cores, following BASELINE.md 4 / SURVEY 8d: BASELINE configs[0] (interm_8m,
    x[2,5,32,64] -> [2,1,128,256], mse) and configs[1] (interm_117m, batch 8, 23 -> 3 variables, 32x64 -> 128x256,
    bayesian_tv), each UN-SCALED: 2 warm-up steps, median of 5 timed steps of fwd + loss + bwd + AdamW.  The headline
    model (interm_1b at 128x256, L = 8192) does not finish on a CPU in the bench's time budget; one 16x32 tile of it
    extrapolated by the FLOP ratio is kept as a separately named estimate."""
    import statistics
    from oracle import orbit2_oracle as O
    cores = _host_threads()
    torch.set_num_threads(cores)
    c1_in = CONST + ["total_precipitation_24hr"]
    t1 = _oracle_steps("interm_8m", c1_in, ["total_precipitation_24hr"], 2, (32, 64), "mse", 2, 5, "config1 interm_8m")
    t2 = _oracle_steps("interm_117m", ERA5_VARS, OUT_VARS, 8, (32, 64), "bayesian_tv", 2, 5, "config2 interm_117m")
    m1, m2 = statistics.median(t1), statistics.median(t2)
    out = {"value": 8.0 / m2, "unit": "samples/s", "cores": cores, "kind": "port",
           # `value` is NOT the GPU line's workload (interm_1b at 128x256 cannot finish on host cores in the bench's
           # budget): a reader must not divide one by the other; the same-model estimate is `interm_1b_tile_estimate`
           "same_workload": False,
           "sample": "NOT the workload of the GPU line (interm_1b, 128x256): that one is only ESTIMATED, see "
                     "`interm_1b_tile_estimate` (one 16x32 tile extrapolated by the FLOP ratio).  `value` here = "
                     "oracle (plain PyTorch fp32 restatement of the reference) fwd+loss+bwd+AdamW of BASELINE configs[1]: "
                     "interm_117m, x[8,23,32,64] -> [8,3,128,256], bayesian_tv; 2 warm-up steps, median of 5 timed steps "
                     "(%.2f s/step); measured, not extrapolated.  The GPU line above is interm_1b at 128x256, which no CPU "
                     "run of this length can reach: see `interm_1b_tile_estimate`" % m2,
           "config1_interm_8m": {"value": 2.0 / m1, "unit": "samples/s", "ms_per_step": 1e3 * m1,
                                 "sample": "x[2,5,32,64] -> [2,1,128,256], mse, fp32; 2 warm-ups, median of 5"},
           "config2_interm_117m": {"value": 8.0 / m2, "unit": "samples/s", "ms_per_step": 1e3 * m2,
                                   "sample": "x[8,23,32,64] -> [8,3,128,256], bayesian_tv, fp32; 2 warm-ups, median of 5"}}
    if model_name == "interm_1b":
        grid = (16, 32)
        tt = _oracle_steps(model_name, ERA5_VARS, OUT_VARS, 1, grid, "bayesian_tv", 1, 2, "interm_1b 16x32 tile")
        dt = sum(tt) / len(tt)
        m = MODELS[model_name]
        f_tile = forward_flops(grid[0] * grid[1] // 4, V, m["embed_dim"], m["depth"], 4, C, grid[0], grid[1], m["num_heads"])
        f_full = forward_flops(8192, V, m["embed_dim"], m["depth"], 4, C, 128, 256, m["num_heads"])
        out["interm_1b_tile_estimate"] = {
            "value": (1.0 / dt) * f_tile / f_full, "unit": "samples/s",
            "sample": "ESTIMATE: interm_1b on one 16x32 tile (L=128), batch 1, 2 timed steps of %.2f s, extrapolated to the "
                      "128x256 grid by the dense-FLOP ratio %.5f (attention is ~0 %% of the tile's FLOPs and 22 %% of "
                      "the full grid's, so this flatters the CPU)" % (dt, f_tile / f_full)}
    return out


def forward_flops(L, V, D, depth, dd, C, h, w, heads, p=2, s=4, cr=4, r=4, folded_varagg=False):
    """Forward FLOPs per sample, SURVEY.md 8(d) (multiply-add = 2; softmax / LN / GELU not counted); model FLOPs = 3x this.
    folded_varagg=True: what this build executes for the variable aggregation (5 MACs per (token, variable, channel) + the
    scores + the projection) instead of the reference's dense kv GEMM.  tests/test_oracle_golden.py pins it to the survey's
    probe values and to the oracle's copy."""
    pe = 2 * L * V * p * p * D
    if folded_varagg:
        va = 2 * L * V * D * 5 + 2 * L * V * heads * 5 + 2 * L * D * D
    else:
        va = 2 * L * V * D * 2 * D + 2 * (2 * L * D * D) + 4 * L * V * D + pe
    blk = depth * (2 * L * D * 3 * D + 4 * L * L * D + 2 * L * D * D + 4 * L * D * r * D)
    head = dd * 2 * L * D * D + 2 * L * D * C * (p * s) ** 2
    convs = 2 * h * w * (C + 4) * (cr * s * s) * 9 + 2 * (16 * h * w) * cr * C * 9 + 2 * (16 * h * w) * C * C * 9
    return va + blk + head + convs


OTHER_CONFIGS = (
    # (key, BASELINE.json config it stands for, bench arguments, seconds allowed)
    ("interm_117m_32x64_b8", "configs[1]: interm_117m bf16, ERA5 5.625 -> 1.40625 deg synthetic grids, 1 x MI355X",
     ["--model", "interm_117m", "--grid", "32x64", "--batch", "8", "--steps", "40", "--warmup", "10"], 150),
    ("interm_1b_daymet_96x192_b4", "configs[4]: interm_1b, Daymet-like 7 -> 3 variables, hybrid perceptual + lat-weighted MSE loss",
     ["--daymet", "--grid", "96x192", "--batch", "4", "--steps", "6", "--warmup", "2"], 150),
    ("interm_10b_128x256_b2", "configs[3]: interm_10b bf16, ERA5 1.40625 -> 0.25 deg (128x256), per-GPU batch 2",
     ["--model", "interm_10b", "--batch", "2", "--steps", "3", "--warmup", "1"], 240),
)


def write_npz_tree(root, hw, variables, n_files, per_file, seed):
    """a directory tree in the reference's on-disk format (data/processing/nc2npz.py:22-166 writes it there; reader:
    data/iterdataset.py:46-177): <root>/train/<year>_<shard>.npz with var -> float32 [T, 1, H, W], lat.npy, lon.npy,
    normalize_{mean,std}.npz, train/climatology.npz.  Values are the bench's synthetic recipe in physical units."""
    rng = np.random.default_rng(seed)
    H, W = hw
    os.makedirs(os.path.join(root, "train"), exist_ok=True)
    for f in range(n_files):
        arrs = {}
        for v in variables:
            a = rng.standard_normal((per_file, 1, H, W), dtype=np.float32)
            arrs[v] = (np.abs(a) * 1e-3) if "precip" in v else (a + 270.0)
        np.savez(os.path.join(root, "train", "2000_%d.npz" % f), **arrs)
    np.savez(os.path.join(root, "train", "climatology.npz"), **{v: np.zeros((1, H, W), np.float32) for v in variables})
    np.save(os.path.join(root, "lat.npy"), np.linspace(-90.0, 90.0, H))
    np.save(os.path.join(root, "lon.npy"), np.linspace(0.0, 360.0, W, endpoint=False))
    np.savez(os.path.join(root, "normalize_mean.npz"), **{v: np.array([0.5e-3 if "precip" in v else 270.0]) for v in variables})
    np.savez(os.path.join(root, "normalize_std.npz"), **{v: np.array([1e-3 if "precip" in v else 1.0]) for v in variables})


def npz_leg(a, step_on, fence, in_vars, out_vars, lo_hw, hi_hw, B, resident_sps, dev):
    """the step at the headline configuration fed END TO END through the npz data plane.  `step_on(batch, i)` runs
    one full training step on a loader batch.  Reports samples/s beside the resident-batch figure, the loader's CPU seconds
    per sample (one process, no GPU work: what a worker spends reading, normalising and collating one sample) and how long the
    consumer waited for batches."""
    import shutil
    import tempfile
    import climate_learn as cl
    root = tempfile.mkdtemp(prefix="orbit2_npz_", dir=os.environ.get("ORBIT2_NPZ_DIR", "/tmp"))
    try:
        # one batch per shard file; every worker owns the same number of files and the whole measurement (warm-up + timed steps)
        # stays inside ONE pass over the tree: a pass change tears the worker pool down and forks a new one from a process that
        # maps > 100 GB of device memory (seconds; measured 13 s with 8 files and 10 batches) -- real trees hold years of shards per
        # pass, so that cost is per epoch, not per few steps, and is not what this leg is about
        wk = max(1, a.data_workers)
        per_file, n_files = B, -(-(max(1, a.warmup) + a.steps + 2) // wk) * wk
        t0 = time.perf_counter()
        write_npz_tree(os.path.join(root, "lo"), lo_hw, in_vars, n_files, per_file, 1)
        write_npz_tree(os.path.join(root, "hi"), hi_hw, out_vars, n_files, per_file, 2)
        t_write = time.perf_counter() - t0
        pin = os.environ.get("ORBIT2_NPZ_PIN", "1") == "1"
        mk = lambda workers: cl.data.IterDataModule("downscaling", os.path.join(root, "lo"), os.path.join(root, "hi"), in_vars, out_vars,
                                                    batch_size=B, buffer_size=0, num_workers=workers, pin_memory=pin)
        dm0 = mk(0)
        dm0.setup()
        t0 = time.perf_counter()
        n0 = 0
        for bt in dm0.train_dataloader():                            # loader cost alone, one process
            n0 += bt[0].shape[0]
            if n0 >= 2 * B:
                break
        cpu_s_per_sample = (time.perf_counter() - t0) / max(n0, 1)
        dm = mk(a.data_workers)
        dm.setup()

        def batches():
            while True:                                              # epochs back to back (a fresh loader per epoch, as the driver does)
                for bt in dm.train_dataloader():
                    if bt[0].shape[0] == B:
                        yield bt
        it = batches()
        for i in range(max(1, a.warmup)):
            step_on(next(it), i)
        fence()
        waited = queued = 0.0
        t0 = time.perf_counter()
        for i in range(a.steps):
            tw = time.perf_counter()
            bt = next(it)
            ts = time.perf_counter()
            waited += ts - tw
            step_on(bt, a.warmup + i)
            queued += time.perf_counter() - ts
        fence()
        dt = time.perf_counter() - t0
        sps = B * a.steps / dt
        return {"value": sps, "unit": "samples/s", "ms_per_step": 1e3 * dt / a.steps, "steps": a.steps,
                "resident_value": resident_sps, "ratio_to_resident": sps / resident_sps,
                "workers": a.data_workers, "pin_memory": pin, "host_ms_per_step_in_the_step_call": 1e3 * queued / a.steps, "upload": "non_blocking .to(device) of pinned batches (trainer.training_step)",
                "loader_cpu_s_per_sample": cpu_s_per_sample,
                "loader_cores_needed_at_this_rate": cpu_s_per_sample * sps,
                "consumer_wait_ms_per_step": 1e3 * waited / a.steps,
                "tree": {"format": "reference npz tree (nc2npz.py layout), float32", "low_res": list(lo_hw), "high_res": list(hi_hw),
                         "files": n_files, "samples_per_file": per_file, "write_s": t_write,
                         "bytes_per_sample": 4 * (len(in_vars) * lo_hw[0] * lo_hw[1] + len(out_vars) * hi_hw[0] * hi_hw[1])},
                "path": "IterDataModule(inp_root_dir, out_root_dir).train_dataloader(): NpyReader file sharding -> Downscale -> "
                        "Normalize / LogTransform -> collate_fn (data/iterdataset.py, itermodule.py)"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def other_configs(t_start, budget_s=420.0, configs=None):
    """The other BASELINE configs on the driver's clock: each one is this same script run as a CHILD process
    (a fresh interpreter started with subprocess after this process has released its device memory; never an exec of the
    process that holds the GPU), its JSON line reduced to {value, ms_per_step, roofline.frac, config}.  A config is skipped,
    and says so, when the time already spent plus its allowance would pass `budget_s` (the default run must stay within minutes)."""
    import subprocess
    out = {}
    for key, what, args, allow in (configs or OTHER_CONFIGS):
        spent = time.perf_counter() - t_start
        if spent + allow > budget_s:
            out[key] = {"skipped": "time budget: %.0f s spent, %d s allowance, %.0f s budget" % (spent, allow, budget_s), "stands_for": what}
            continue
        cmd = [sys.executable, os.path.abspath(__file__)] + args + ["--no-cpu-baseline", "--no-other-configs"]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=allow)
            lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[key] = {"error": "exit code %d" % r.returncode, "stderr_tail": r.stderr.decode(errors="replace")[-600:], "stands_for": what}
                continue
            d = json.loads(lines[-1])
            ra = d.get("roofline_attention") or {}
            out[key] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
                        "roofline": {"frac": (d.get("roofline") or {}).get("frac"), "bound": "mfma",
                                     "note": "executed FLOPs of the step / 2.5 PFLOP/s, as the headline line"},
                        "roofline_gemm_frac": (d.get("roofline_gemm") or {}).get("frac"),
                        "roofline_attention_frac": ra.get("frac"),
                        "config": d["config"], "dtype": d["dtype"], "stands_for": what, "wall_s": time.perf_counter() - t0,
                        "command": "python bench.py " + " ".join(args)}
        except subprocess.TimeoutExpired:
            out[key] = {"error": "still running after %d s (killed)" % allow, "stands_for": what}
    return out


def self_launch(a):
    """`python bench.py --gpus N` from a plain shell (no RANK in the environment): start the N ranks ourselves, one per GPU,
    through torch.distributed.run -- as a CHILD process and before this process has made any GPU call (replacing a
    process that has initialised the GPU is not allowed on this pool) -- relay rank 0's single JSON line, and exit with the
    children's return code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    sys.exit(r.returncode if r.returncode else (0 if lines else 1))


def launch_check(a, rank, world, local):
    """--launch-check: rendezvous only (gloo, no GPU call, no workload): proves that N ranks were started, met, and that
    rank 0 prints one line carrying n_gpus = N.  `value` is null: this is not a measurement."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        devs = [None] * world
        dist.all_gather_object(devs, local)
        dist.barrier()
        ranks = dist.get_world_size()
        dist.destroy_process_group()
    else:
        devs, ranks = [local], 1
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "samples/s", "n_gpus": world, "steps": 0, "warmup": 0,
                          "launch_check": True, "ranks_met": ranks, "backend": "gloo", "devices": devs}), flush=True)


def _attn_roofline(prof, traffic_attn, note):
    f, b = prof.get("attn_fwd"), prof.get("attn_bwd")
    if not f or not b or not f["launches"]:
        return None
    work, ms, n = f["work"] + b["work"], f["ms"] + b["ms"], f["launches"] + b["launches"]
    ach = work / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": ach / (PEAK_BF16 / 1e12),
            "forward": {"achieved": f["work"] / f["ms"] / 1e9, "frac": f["work"] / f["ms"] / 1e9 / (PEAK_BF16 / 1e12),
                        "avg_launch_ms": f["ms"] / f["launches"], "algorithmic_bytes_per_launch": f["bytes"]},
            "backward": {"achieved": b["work"] / b["ms"] / 1e9, "frac": b["work"] / b["ms"] / 1e9 / (PEAK_BF16 / 1e12),
                         "avg_launch_ms": b["ms"] / b["launches"], "algorithmic_bytes_per_launch": b["bytes"],
                         "note": "one API launch = delta + dQ + fused dK/dV kernels; 7 matrix products executed for 4 credited"},
            "traffic": traffic_attn, "traffic_note": note,
            "algorithmic_bytes_per_launch": (f["bytes"] * f["launches"] + b["bytes"] * b["launches"]) / n,
            "kernel": "orbit2_attn_fwd / orbit2_attn_bwd (csrc/attn.hip): every attention launch of the timed region",
            "launches": n, "avg_launch_ms": ms / n}


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        self_launch(a)
    if a.launch_check:
        return launch_check(a, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                            int(os.environ.get("LOCAL_RANK", "0")))
    # stdout carries exactly ONE line, the result: libraries that write banners to file descriptor 1 (RCCL prints its
    # version block there at communicator creation) are sent to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (plain `python bench.py --gpus N` does it)"
                         % (a.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force = os.environ.get("ORBIT2_FORCE_COLLECTIVES", "0") == "1"
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("ORBIT2_DIST_BACKEND", "nccl")     # gloo: multi-rank rehearsal on one card (tests)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import climate_learn as cl
    from climate_learn import _hip
    from climate_learn.metrics import Bayesian_TV
    from climate_learn.models.hub import Res_Slim_ViT
    from climate_learn.models.hub.components.vit_blocks import Block
    from climate_learn.trainer import training_step
    if a.no_fused_colsum:
        from climate_learn import _ops
        _ops._FUSE_COLSUM = False

    m = MODELS[a.model]
    h, w = (int(v) for v in a.grid.split("x"))
    in_vars = (CONST + OUT_VARS) if a.daymet else ERA5_VARS
    V, C, B = len(in_vars), len(OUT_VARS), a.batch
    L = h * w // 4
    drop = 0.0 if a.no_dropout else 0.1
    tp = a.tensor_par
    capturable = world == 1 or os.environ.get("ORBIT2_DIST_BACKEND", "nccl") == "nccl"     # gloo rehearsals cannot be captured
    # (the parameter-sharding engine is captured on request only -- `--fsdp --graph on` -- in its single-stream form: DESIGN 5)
    a.graph = a.graph == "on" or (a.graph == "auto" and B * L <= 16384 and tp == 1 and capturable and not a.fsdp)
    if tp > 1 and (world % tp or a.graph):
        raise SystemExit("--tensor-par %d needs WORLD_SIZE divisible by it and no --graph" % tp)
    dp_world, dp_rank = world // tp, rank // tp
    dp_group = tp_group = None
    if tp > 1:       # reference rank layout (examples/intermediate_downscaling.py:173-247): tp ranks adjacent
        for i in range(dp_world):
            g_ = dist.new_group(list(range(i * tp, (i + 1) * tp)))
            tp_group = g_ if rank // tp == i else tp_group
        for i in range(tp):
            g_ = dist.new_group([i + j * tp for j in range(dp_world)])
            dp_group = g_ if rank % tp == i else dp_group
    torch.manual_seed(0)
    cl.manual_seed(0, dp_rank)
    with torch.device(dev):
        model = Res_Slim_ViT(in_vars, (h, w), V, C, 1, superres_mag=4, cnn_ratio=4, patch_size=2, drop_path=drop,
                             drop_rate=drop, learn_pos_emb=True, embed_dim=m["embed_dim"], depth=m["depth"],
                             decoder_depth=4, num_heads=m["num_heads"], mlp_ratio=4, FusedAttn_option=cl.FusedAttn.HIP,
                             tensor_par_size=tp, tensor_par_group=tp_group)
    if tp > 1:
        cl.dist.tp.sync_replicated(model, tp_group)
    model.data_config(156.0, (h, w), V, C)
    for blk in model.blocks:
        blk.recompute = a.recompute
    nparams = sum(p.numel() for p in model.parameters())
    if a.fsdp:
        eng = cl.HipFullyShardedDataParallel(model, process_group=dp_group, unit_types=(Block, nn.Sequential), tp_group=tp_group)
    else:
        eng = cl.HipDataParallel(model, process_group=dp_group, unit_types=(Block, nn.Sequential),
                                 shard_optimizer=a.shard_optimizer, replica_group=tp_group)
    opt = cl.load_optimizer(eng, "adamw", {"lr": 5e-4, "betas": (0.9, 0.99), "weight_decay": 1e-5})
    scaler = cl.HipGradScaler(init_scale=8192.0, growth_interval=100, min_scale=128.0, sync_world=tp > 1)
    loss_fn = Bayesian_TV(aggregate_only=True)
    if a.daymet:
        os.environ.setdefault("ORBIT2_LPIPS_SYNTHETIC", "1")    # throughput run: seeded stand-in LPIPS-VGG16 weights, opt-in
        # BASELINE configs[4] / SURVEY 8d-5: "hybrid perceptual + lat-weighted MSE": L1 + 0.5 LPIPS-VGG16 + intended lat_mse
        from climate_learn.metrics.utils import MetricsMetaInfo
        hy_ = (721 if (h, w) == (128, 256) else 4 * h)
        meta = MetricsMetaInfo(in_vars, OUT_VARS, np.linspace(-90.0, 90.0, hy_), None, None)
        loss_fn = cl.load_loss(dev, None, a.daymet_loss, True, meta)
    eng.train()

    # synthetic ERA5-shaped batch, resident in HBM before the timed region (SURVEY 8d input recipe)
    g = torch.Generator().manual_seed(1000 + dp_rank)
    x = torch.randn(B, V, h, w, generator=g)
    gc = torch.Generator().manual_seed(7)
    for i in range(4):
        x[:, i] = torch.randn(h, w, generator=gc)
    hy, wy = (721, 1440) if (h, w) == (128, 256) else (4 * h, 4 * w)
    y = torch.randn(B, C, hy, wy, generator=g)
    y[:, 0] = torch.log1p(torch.relu(y[:, 0]))
    batch = (x.to(dev), y.to(dev), in_vars, OUT_VARS)

    def step(i):
        loss = training_step(batch, i, eng, dev, VAR_WEIGHTS, loss_fn)
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1 or force:
            dist.barrier()
        torch.cuda.synchronize()

    if a.mall_probe:
        _hip.mall_calibration()
    for i in range(a.warmup):
        step(i)
    fence()
    if (world > 1 or force) and not a.graph and not a.no_comm_stats:
        eng.comm_stats = cl.CommStats()                 # HIP-event accounting of the collectives over the timed region
    if a.graph:
        # instrumented eager pass for the per-kernel (roofline) numbers, then the captured step for the throughput
        _hip.timer = _hip.KernelTimer()
        for i in range(min(3, a.steps)):
            step(a.warmup + i)
        fence()
        prof = _hip.timer.summary()
        for rec in prof.values():                      # scale the instrumented pass to a.steps steps
            rec["work"] *= a.steps / min(3, a.steps)
            rec["ms"] *= a.steps / min(3, a.steps)
            rec["launches"] = int(rec["launches"] * a.steps / min(3, a.steps))
        _hip.timer = None
        gstep = cl.GraphedTrainStep(eng, loss_fn, batch, VAR_WEIGHTS, scaler=scaler)

        def step(i):                                   # noqa: F811  (graphed fwd+bwd, eager scaler + AdamW)
            loss = gstep()
            scaler.step(opt)
            scaler.update()
            return loss

        for i in range(max(1, a.warmup)):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(a.steps):
            last = step(a.warmup + i)
        fence()
        dt = time.perf_counter() - t0
    else:
        _hip.timer = _hip.KernelTimer()
        t0 = time.perf_counter()
        for i in range(a.steps):
            last = step(a.warmup + i)
        fence()
        dt = time.perf_counter() - t0
        prof = _hip.timer.summary()
        _hip.timer = None
    tmax = torch.tensor([dt], device=dev)
    devices = [torch.cuda.current_device()]
    backend_name, ranks_met = None, 1
    if world > 1 or force:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        devices = [None] * world
        dist.all_gather_object(devices, torch.cuda.current_device())
        backend_name, ranks_met = dist.get_backend(), dist.get_world_size()
    dt = float(tmax.item())
    loss_val = float(last.detach())

    if rank == 0:
        sps = dp_world * B * a.steps / dt
        f_dense = forward_flops(L, V, m["embed_dim"], m["depth"], 4, C, h, w, m["num_heads"])
        f_exec = forward_flops(L, V, m["embed_dim"], m["depth"], 4, C, h, w, m["num_heads"], folded_varagg=True)
        gm = prof.get("gemm_bf16", {"work": 0.0, "ms": 1.0, "launches": 0})
        # Counter traffic (FETCH_SIZE x2 + WRITE_SIZE per launch) from the committed PMC passes of tools/profile_round.sh.  The file
        # is stamped with the library's source hash and the configuration: it is used only for THE SAME build and configuration
        # (a stale file gives null, never a silently wrong number).
        traffic, traffic_attn, traffic_step, traffic_note, tj = None, None, None, None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_JSON)))
            here = open(_hip.LIB_PATH + ".srchash").read().strip() if os.path.exists(_hip.LIB_PATH + ".srchash") else None
            if tj["_config"] != {"model": a.model, "batch": B, "grid": a.grid}:
                traffic_note = "profiles/%s is for another configuration" % TRAFFIC_JSON
            elif tj.get("_srchash") != here:
                traffic_note = "profiles/%s was collected on another build of the library (srchash differs)" % TRAFFIC_JSON
            else:
                traffic = tj["gemm"]["hbm_bytes_per_launch"]
                traffic_step = tj.get("step", {}).get("hbm_bytes_per_step")
                traffic_attn = {k: tj[k]["hbm_bytes_per_launch"] for k in ("attn_fwd", "attn_bwd_dq", "attn_bwd_dkv") if k in tj}
        except Exception as e:
            traffic_note = "no usable profiles/%s (%s)" % (TRAFFIC_JSON, type(e).__name__)
        split = None       # Infinity-Cache hits vs HBM reads inside that counter traffic (mean L2-miss latency of the step's kernels)
        try:
            mj = json.load(open(os.path.join(ROOT, "profiles", MALL_JSON)))
            share = {k: v["infinity_cache_hit_share_est"] for k, v in mj.items() if isinstance(v, dict) and "gemm256" in k
                     and "infinity_cache_hit_share_est" in v}
            if share and traffic is not None:
                pick = lambda tag, grouped: [v for k, v in share.items() if tag in k and ("grouped" in k) == grouped]
                fam = {f: min(v) for f, v in (("gemm_8phase_single", pick("gemm256t", False)), ("gemm_8phase_grouped_dw", pick("gemm256t", True)),
                                              ("gemm_w4_single", pick("gemm256w", False)), ("gemm_w4_grouped_dw", pick("gemm256w", True))) if v}
                num = den = 0.0
                for f, sh in fam.items():
                    if f in tj:                      # per-family counter traffic x (1 - Infinity-Cache share of its reads)
                        n = tj[f]["launches"]
                        num += n * (tj[f]["fetch_bytes_per_launch"] * (1.0 - sh) + tj[f]["write_bytes_per_launch"])
                        den += n
                split = {"infinity_cache_hit_share_of_reads": share,
                         "hbm_bytes_per_launch_estimate": (num / den) if den else None,    # NOT floored at the algorithmic bytes
                         "calibration_cycles": mj.get("_calibration"),
                         "method": "mean L2-miss latency (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) of the kernels INSIDE the bench step, "
                                   "interpolated between an Infinity-Cache-hit stream and an HBM stream measured lightly loaded and "
                                   "saturating (the lower share is used), per kernel family; writes counted in full; profiles/"
                                   + MALL_JSON}
        except Exception:
            split = None
        ach = gm["work"] / (gm["ms"] * 1e-3) / 1e12 if gm["launches"] else 0.0
        out = {
            "metric": METRIC, "value": sps, "unit": "samples/s", "n_gpus": world, "rccl_ranks": ranks_met,
            "backend": backend_name, "devices": devices, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s Res_Slim_ViT bf16 (fp32 master), %s synthetic: x[%d,%d,%d,%d] "
                                   "-> pred[%d,%d,%d,%d], target %dx%d cropped; fwd+%s loss+bwd+grad all-reduce+"
                                   "loss-scaled fused AdamW; dropout %.1f, drop-path %.1f"
                                   % (a.model, "Daymet-like multi-variable" if a.daymet else "ERA5 1.40625deg->0.25deg",
                                      B, V, h, w, B, C, 4 * h, 4 * w, hy, wy,
                                      (a.daymet_loss + " (L1 + 0.5 LPIPS-VGG16" + (" + lat-weighted MSE)" if "lat" in a.daymet_loss else ")")) if a.daymet else "bayesian_tv", drop, drop),
                       "per_gpu_batch": B, "global_batch": B * dp_world, "tokens_per_sample": L, "params": nparams,
                       "parallelism": (("fsdp%d" if a.fsdp else "dp%d") % world) if tp == 1 else "dp%dxtp%d" % (dp_world, tp),
                       "activation_recompute": bool(a.recompute),
                       "hipgraph": bool(a.graph),
                       "loss": a.daymet_loss if a.daymet else "bayesian_tv", "in_vars": V},
            # THE STEP against the MFMA peak: FLOPs the kernels of one step EXECUTE (variable aggregation folded, SURVEY 8d: "subtract
            # the saved FLOPs ... report which formulation was executed"; attention-backward recomputation NOT credited) x samples/s
            # per GPU.  One "launch" = one step (fwd + loss + bwd + grad all-reduce + loss-scaled AdamW, everything inside the timed
            # region); traffic = counter bytes of ALL kernels of a step.  The per-family figures follow in roofline_gemm /
            # roofline_attention; the dense-formulation figure (crediting the folded FLOPs) is context in step_model.
            "roofline": {"bound": "mfma", "achieved": 3 * f_exec * sps / world / 1e12, "peak": PEAK_BF16 / 1e12,
                         "unit": "TFLOP/s", "frac": 3 * f_exec * sps / world / PEAK_BF16,
                         "traffic": traffic_step, "traffic_note": traffic_note,
                         "algorithmic_flops_per_launch": 3 * f_exec * B,
                         "kernel": "the whole training step: every kernel of the timed region (csrc/*.hip), executed-FLOP formulation "
                                   "(folded variable aggregation); one launch = one step of %d samples" % B,
                         "launches": a.steps, "avg_launch_ms": 1e3 * dt / a.steps},
            # every GEMM of the step (the family that holds most of its FLOPs), same fields, per kernel launch
            "roofline_gemm": {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                              "frac": ach / (PEAK_BF16 / 1e12), "traffic": traffic, "traffic_split": split,
                              "algorithmic_bytes_per_launch": gm.get("bytes", None),
                              "kernel": "orbit2_gemm_bf16 / orbit2_gemm_bf16_grouped (csrc/gemm.hip, hand-written MFMA kernels): "
                                        "EVERY GEMM of the step -- forward, input gradients, weight gradients -- all launches "
                                        "of the timed region; no vendor-library GEMM is called",
                              "traffic_note": traffic_note,
                              "launches": gm["launches"],
                              "avg_launch_ms": gm["ms"] / max(1, gm["launches"])},
            # the family FURTHEST below its roofline, with the same fields: forward and backward launches of the
            # flash-attention kernels (csrc/attn.hip); algorithmic FLOPs 4 B H L^2 d forward, 8 B H L^2 d backward (the backward's
            # recomputation of S and its two-kernel split are NOT credited)
            "roofline_attention": _attn_roofline(prof, traffic_attn, traffic_note),
            "step_model": {
                "model_flops_per_sample_dense": 3 * f_dense, "executed_flops_per_sample_folded_varagg": 3 * f_exec,
                "mfma_frac_of_peak_dense_formulation": 3 * f_dense * sps / world / PEAK_BF16,
                "mfma_frac_of_peak_executed": 3 * f_exec * sps / world / PEAK_BF16,
                "attn_fwd_tflops": (prof["attn_fwd"]["work"] / prof["attn_fwd"]["ms"] / 1e9) if "attn_fwd" in prof else None,
                "attn_bwd_tflops": (prof["attn_bwd"]["work"] / prof["attn_bwd"]["ms"] / 1e9) if "attn_bwd" in prof else None,
                "gemm_ms_per_step": gm["ms"] / a.steps,
                "library_gemm": {"provider": "none (the torch.matmul / hipBLASLt dispatch of round 1 is deleted)",
                                 "launches": 0, "ms_per_step": 0.0, "share_of_gemm_flops": 0.0},
                "attn_ms_per_step": (prof.get("attn_fwd", {"ms": 0})["ms"] + prof.get("attn_bwd", {"ms": 0})["ms"]) / a.steps,
                "final_loss": loss_val, "loss_scale": scaler.get_scale()},
        }
        if eng.comm_stats is not None:                   # did the collectives hide behind backward?
            cst = eng.comm_stats.summary(a.steps)
            out["comm"] = dict(cst, note="HIP events: comm_ms = busy time of the communication stream in the bucket collectives "
                                         "(+ unit all-gathers with --fsdp), exposed_comm_ms = what the compute stream waited for "
                                         "them at the join / at a gathered unit's hand-over; bytes = buffers handed to the collectives",
                               overlap_fraction=(1.0 - cst["exposed_comm_ms_per_step"] / cst["comm_ms_per_step"])
                               if cst["comm_ms_per_step"] > 0 else None)
            out["comm_ms_per_step"] = cst["comm_ms_per_step"]
            out["exposed_comm_ms"] = cst["exposed_comm_ms_per_step"]
        if a.data == "npz" and world == 1 and not a.graph:
            def step_on(bt, i):
                loss = training_step(bt, i, eng, dev, VAR_WEIGHTS, loss_fn)
                opt.zero_grad()
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
            out["data_npz"] = npz_leg(a, step_on, fence, in_vars, OUT_VARS, (h, w), (hy, wy), B, sps, dev)
            out["data"] = "synthetic, resident in HBM (`value`); `data_npz` = the same step fed from a reference-format npz tree"
        default_run = (a.model == "interm_1b" and a.grid == "128x256" and not a.daymet and tp == 1 and not a.fsdp and not a.recompute)
        if world == 1 and (a.eager_baseline or not a.no_cpu_baseline or (default_run and not a.no_other_configs) or a.other_configs_smoke):
            # release the headline run's device memory: the closures hold the engine, the optimizer and the batch too
            import gc as _gc
            step = fence = last = gstep = loss_fn = scaler = x = y = None          # noqa: F841
            del eng, opt, model, batch
            _gc.collect()
            torch.cuda.empty_cache()
            print("[bench] released; device memory still allocated: %.1f GB" % (torch.cuda.memory_allocated() / 1e9), file=sys.stderr)
        if world == 1 and a.other_configs_smoke:
            out["other_configs"] = other_configs(T_PROCESS_START, configs=(
                ("interm_8m_32x64_b2", "configs[0]-shaped smoke entry (tests)", ["--model", "interm_8m", "--grid", "32x64", "--batch", "2",
                                                                                  "--steps", "3", "--warmup", "1"], 240),))
        elif world == 1 and default_run and not a.no_other_configs:
            out["other_configs"] = other_configs(T_PROCESS_START)
        if world == 1 and a.eager_baseline and not a.daymet:
            out["gpu_eager_baseline"] = gpu_eager_baseline(a.model, V, C, a.eager_baseline, dev)
            torch.cuda.empty_cache()
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.model, V, C)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
