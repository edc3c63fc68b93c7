"""Synthetic stand-in for the reference's IterDataModule (data/itermodule.py:29-231,385-469): same constructor
keywords and accessor surface, yielding seeded synthetic (x, y, in_variables, out_variables) batches of the
configured grid sizes.  The npz shard reader / tiling / normalisation data plane is SURVEY 8(f)-1 (next).

Rank sharding follows the reference's intent (iterdataset.py:68-88: each data-parallel rank reads its own
files): rank r draws from seed base + r."""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

_CONST = ("land_sea_mask", "orography", "lattitude", "landcover")


class SyntheticGridDataModule:
    def __init__(self, in_vars: Sequence[str], out_vars: Sequence[str], lowres_hw, highres_hw, batch_size=2,
                 steps_per_epoch=4, seed=0, rank=0, device="cpu"):
        self.in_vars, self.out_vars = list(in_vars), list(out_vars)
        self.lo, self.hi = tuple(lowres_hw), tuple(highres_hw)
        self.batch_size, self.steps = batch_size, steps_per_epoch
        self.seed, self.rank = seed, rank
        self.device = device
        self._ready = False

    # ---- reference surface ----------------------------------------------------------------------------
    def to(self, device):
        self.device = device
        return self

    def setup(self, stage=None):
        self.lat = np.linspace(-90.0, 90.0, self.hi[0])
        self.lon = np.linspace(0.0, 360.0, self.hi[1], endpoint=False)
        g = torch.Generator().manual_seed(self.seed)          # constants are identical on every rank
        self._consts = {v: torch.randn(self.lo, generator=g) for v in _CONST if v in self.in_vars}
        self._ready = True

    def get_lat_lon(self):
        return (self.lat, self.lon) if self._ready else (None, None)

    def get_data_dims(self):
        return (torch.Size([self.batch_size, len(self.in_vars), *self.lo]),
                torch.Size([self.batch_size, len(self.out_vars), *self.hi]))

    def get_data_variables(self):
        return self.in_vars, self.out_vars

    def get_climatology(self, split="val"):
        return {v: torch.zeros(self.hi) for v in self.out_vars}

    def get_out_transforms(self):
        return {v: SimpleNamespace(mean=0.0, std=1.0) for v in self.out_vars}

    def _batch(self, g):
        B = self.batch_size
        x = torch.randn(B, len(self.in_vars), *self.lo, generator=g)
        for i, v in enumerate(self.in_vars):
            if v in self._consts:
                x[:, i] = self._consts[v]
        y = torch.randn(B, len(self.out_vars), *self.hi, generator=g)
        if "total_precipitation_24hr" in self.out_vars:
            i = self.out_vars.index("total_precipitation_24hr")
            y[:, i] = torch.log1p(torch.relu(y[:, i]))
        return x, y, self.in_vars, self.out_vars

    def train_dataloader(self):
        if not self._ready:
            raise RuntimeError("Data module has not been set up yet.")
        g = torch.Generator().manual_seed(self.seed * 7919 + 1 + self.rank)
        return [self._batch(g) for _ in range(self.steps)]

    def val_dataloader(self):
        g = torch.Generator().manual_seed(self.seed * 7919 + 500009 + self.rank)
        return [self._batch(g) for _ in range(1)]

    test_dataloader = val_dataloader


class IterDataModule(SyntheticGridDataModule):
    """Reference keyword surface (itermodule.py:33-56).

    * `inp_root_dir` / `out_root_dir` that exist on disk -> the npz data plane (climate_learn.data.iterdataset):
      shards <root>/{train,val,test}/*.npz, lat.npy / lon.npy, normalize_{mean,std}.npz, <split>/climatology.npz,
      per-rank x per-worker file sharding, div x div tiling with overlap halo, normalisation, shuffle buffer.
    * otherwise -> seeded synthetic grids of `lowres_hw` / `highres_hw` (tiled with the same arithmetic)."""

    def __init__(self, task="downscaling", inp_root_dir=None, out_root_dir=None, in_vars=None, out_vars=None,
                 data_par_size=1, data_par_group=None, src=None, history=1, window=6, pred_range=6, subsample=1,
                 batch_size=64, buffer_size=10000, num_workers=0, pin_memory=False, div=1, overlap=0,
                 lowres_hw=(32, 64), highres_hw=None, steps_per_epoch=4, seed=0):
        import os
        from . import iterdataset as ID
        if task != "downscaling":
            raise NotImplementedError("only the downscaling task is on the hot path")
        rank = 0
        if data_par_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            rank = torch.distributed.get_rank(group=data_par_group)
        self.inp_root_dir, self.out_root_dir = inp_root_dir, out_root_dir
        self.data_par_size, self.data_par_group = data_par_size, data_par_group
        self.subsample, self.buffer_size, self.num_workers = subsample, buffer_size, num_workers
        self.pin_memory = bool(pin_memory)           # page-locked batches: the trainer's non_blocking uploads then run as DMA
        self.div, self.overlap = div, overlap
        self.on_disk = bool(inp_root_dir) and os.path.isdir(os.path.join(str(inp_root_dir), "train"))
        if self.on_disk:
            lat_lo = np.load(os.path.join(inp_root_dir, "lat.npy"))
            lon_lo = np.load(os.path.join(inp_root_dir, "lon.npy"))
            lat_hi = np.load(os.path.join(out_root_dir, "lat.npy"))
            lon_hi = np.load(os.path.join(out_root_dir, "lon.npy"))
            lowres_hw, highres_hw = (len(lat_lo), len(lon_lo)), (len(lat_hi), len(lon_hi))
            self._full_lat, self._full_lon = lat_hi, lon_hi
        highres_hw = highres_hw or (lowres_hw[0] * 4, lowres_hw[1] * 4)
        lo, hi = ID.tile_dims(lowres_hw[0], lowres_hw[1], highres_hw[0], highres_hw[1], div, overlap)
        super().__init__(in_vars, out_vars or in_vars, lo, hi, batch_size, steps_per_epoch, seed, rank)

    # ---- on-disk mode ---------------------------------------------------------------------------------------
    def _files(self, root, split):
        import glob
        import os
        return sorted(f for f in glob.glob(os.path.join(root, split, "*.npz")) if "climatology" not in f)

    def _normalizers(self, root, variables):
        import os
        from . import iterdataset as ID
        from .processing.era5_constants import PRECIP_VARIABLES
        mean = dict(np.load(os.path.join(root, "normalize_mean.npz")))
        std = dict(np.load(os.path.join(root, "normalize_std.npz")))
        return {v: (ID.LogTransform() if v in PRECIP_VARIABLES else ID.Normalize(mean[v][0], std[v][0])) for v in variables}

    def setup(self, stage=None):
        if not self.on_disk:
            return super().setup(stage)
        self.lat, self.lon = self._full_lat, self._full_lon
        self.transforms = self._normalizers(self.inp_root_dir, self.in_vars)
        self.output_transforms = self._normalizers(self.out_root_dir, self.out_vars)
        self._ready = True

    def get_out_transforms(self):
        return dict(self.output_transforms) if self.on_disk else super().get_out_transforms()

    def get_climatology(self, split="val"):
        if not self.on_disk:
            return super().get_climatology(split)
        import os
        clim = np.load(os.path.join(self.out_root_dir, split, "climatology.npz"))
        return {v: torch.from_numpy(np.squeeze(clim[v].astype(np.float32), axis=0)) for v in self.out_vars}

    def _loader(self, split, shuffle):
        from torch.utils.data import DataLoader
        from . import iterdataset as ID
        rd = ID.NpyReader(self._files(self.inp_root_dir, split), self._files(self.out_root_dir, split), self.in_vars,
                          self.out_vars, data_par_size=self.data_par_size, data_par_group=self.data_par_group,
                          shuffle=shuffle, div=self.div, overlap=self.overlap, seed=self.seed)
        # one loader per epoch (the driver asks for a fresh one every epoch): its number is the epoch of the shuffle streams,
        # also when worker processes (which get copies of the datasets) do the iterating
        calls = self.__dict__.setdefault("_loader_calls", {})
        rd.epoch = calls.get(split, 0)
        calls[split] = rd.epoch + 1
        ds = ID.IndividualDataIter(ID.Downscale(rd), self.transforms, self.output_transforms, subsample=self.subsample)
        if shuffle and self.buffer_size > 0:
            ds = ID.ShuffleIterableDataset(ds, self.buffer_size, seed=self.seed, dp_rank=rd.dp_rank)
            ds.epoch = rd.epoch
        chain = [rd] + ([ds] if isinstance(ds, ID.ShuffleIterableDataset) else [])

        class _EpochLoader(DataLoader):
            """DataLoader that numbers its own passes: with num_workers > 0 the workers iterate COPIES of the datasets, so an
            epoch counter advanced inside their __iter__ never reaches the parent and a re-iterated loader would replay the
            identical order (round-2 advisor).  The parent sets the epoch of the reader and of the shuffle buffer before every
            pass -- the workers of that pass are started afterwards and inherit it; `set_epoch(e)` pins it (resume)."""

            def set_epoch(self, epoch: int):
                self._o2_epoch = int(epoch)

            def __iter__(self):
                e = getattr(self, "_o2_epoch", chain[0].epoch)
                for d in chain:
                    d.epoch = e
                self._o2_epoch = e + 1
                return super().__iter__()

        return _EpochLoader(ds, batch_size=self.batch_size, drop_last=False, num_workers=self.num_workers,
                            collate_fn=ID.collate_fn, pin_memory=self.pin_memory and torch.cuda.is_available(),
                            persistent_workers=False)

    def train_dataloader(self):
        return self._loader("train", True) if self.on_disk else super().train_dataloader()

    def val_dataloader(self):
        return self._loader("val", False) if self.on_disk else super().val_dataloader()

    test_dataloader = val_dataloader
