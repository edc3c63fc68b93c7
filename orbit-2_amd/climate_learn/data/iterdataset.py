"""npz shard data plane of the downscaling path (reference: data/iterdataset.py:21-177,313-404,
data/itermodule.py:202-211,451-469, data/precipmodule.py).

On-disk format (reference nc2npz writer): <root>/<split>/<year>_<shard>.npz with one array per variable shaped
[T, 1, H, W]; lat.npy / lon.npy; normalize_{mean,std}.npz; <split>/climatology.npz.

Pieces: per-rank x per-worker file sharding, spatial tiling with overlap halo (a sample is cut into div x div tiles;
odd overlaps split asymmetrically; the horizontal halo is twice the vertical one), per-variable normalisation
(LogTransform for precipitation), shuffle buffer, collate to (x[B,V,h,w], y[B,C,H,W], in_vars, out_vars).
"""
from __future__ import annotations

import random
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import IterableDataset


def overlap_halo(overlap: int) -> Tuple[int, int, int, int]:
    """(left, right, top, bottom) halo widths in low-res pixels (iterdataset.py:112-120)."""
    half = overlap // 2
    if overlap % 2 == 0:
        return 2 * half, 2 * half, half, half
    return 2 * half, 2 * (half + 1), half, half + 1


def _axis_bounds(n: int, div: int, index: int, lo_halo: int, hi_halo: int, mul: int = 1) -> Tuple[int, int]:
    """Half-open [a, b) bounds of tile `index` of `div` along an axis of length n; every tile has the same
    extent n//div + lo_halo + hi_halo: edge tiles take the missing halo from the inner side."""
    if div == 1:
        return 0, n
    a, b = n // div * index, n // div * (index + 1)
    if index == 0:
        b += lo_halo * mul
    else:
        a -= lo_halo * mul
    if index == div - 1:
        a -= hi_halo * mul
    else:
        b += hi_halo * mul
    return a, b


def tile_slices(yinp: int, xinp: int, yout: int, xout: int, div: int, overlap: int):
    """[(in (y1,y2,x1,x2), out (y1,y2,x1,x2))] in the reference's row-major tile order (iterdataset.py:122-170)."""
    left, right, top, bottom = overlap_halo(overlap)
    hmul, vmul = xout // xinp, yout // yinp
    res = []
    for v in range(div):
        for h in range(div):
            xi = _axis_bounds(xinp, div, h, left, right)
            xo = _axis_bounds(xout, div, h, left, right, hmul)
            yi = _axis_bounds(yinp, div, v, top, bottom)
            yo = _axis_bounds(yout, div, v, top, bottom, vmul)
            res.append(((yi[0], yi[1], xi[0], xi[1]), (yo[0], yo[1], xo[0], xo[1])))
    return res


def tile_dims(in_lat: int, in_lon: int, out_lat: int, out_lon: int, div: int, overlap: int):
    """(h, w) of the low-res tile and (H, W) of the high-res tile (itermodule.py:161-198)."""
    if div == 1:
        return (in_lat, in_lon), (out_lat, out_lon)
    left, right, top, bottom = overlap_halo(overlap)
    return ((in_lat // div + top + bottom, in_lon // div + left + right),
            (out_lat // div + (top + bottom) * (out_lat // in_lat), out_lon // div + (left + right) * (out_lon // in_lon)))


def shard_range(n_files: int, rank: int, data_par_size: int, num_workers: int = 1, worker_id: int = 0):
    """(file-list multiplier, remainder, start, end) of iterdataset.py:55-88: the list is wrapped around when
    there are fewer files than rank x worker shards, then split evenly; the tail files are dropped."""
    total = num_workers * data_par_size
    mult, rem = 1, 0
    if n_files < total:
        mult = total // n_files
        rem = total - n_files * mult
        n_files = n_files * mult + rem
    per = n_files // total
    wid = rank * num_workers + worker_id
    return mult, rem, wid * per, wid * per + per


class NpyReader(IterableDataset):
    def __init__(self, inp_file_list, out_file_list, variables, out_variables, data_par_size: int = 1,
                 data_par_group=None, shuffle=False, div=1, overlap=4, rank: Optional[int] = None, seed: int = 0):
        super().__init__()
        # The file order must be IDENTICAL on every rank (the [start:end] slices of shard_range are disjoint only then):
        # the reference gets that from random.seed(0) on every rank; here the reader owns a private stream keyed by
        # (seed, epoch) and never touches Python's global `random`.
        self.seed, self.epoch = int(seed), 0
        assert len(inp_file_list) == len(out_file_list)
        self.inp_file_list = [f for f in inp_file_list if "climatology" not in f]
        self.out_file_list = [f for f in out_file_list if "climatology" not in f]
        self.variables = variables
        self.out_variables = out_variables if out_variables is not None else variables
        self.shuffle, self.div, self.overlap = shuffle, div, overlap
        self.data_par_size, self.data_par_group = data_par_size, data_par_group
        self._rank = rank

    def dp_rank(self) -> int:
        """rank inside the DATA-parallel group: the ranks of one tensor-parallel group share it, so they read the same
        files and (ShuffleIterableDataset) draw the same shuffle-buffer stream"""
        if self._rank is not None:
            return self._rank
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_rank(group=self.data_par_group)
        return 0

    def _my_files(self):
        inp, out = list(self.inp_file_list), list(self.out_file_list)
        if self.shuffle:
            order = list(range(len(inp)))
            random.Random(self.seed * 1000003 + self.epoch).shuffle(order)      # same permutation on every rank
            inp, out = [inp[i] for i in order], [out[i] for i in order]
        wi = torch.utils.data.get_worker_info()
        nw, wid = (wi.num_workers, wi.id) if wi is not None else (1, 0)
        rank = self.dp_rank()
        mult, rem, a, b = shard_range(len(inp), rank, self.data_par_size, nw, wid)
        inp, out = inp * mult + inp[:rem], out * mult + out[:rem]
        return inp[a:b], out[a:b]

    def __iter__(self):
        inp_files, out_files = self._my_files()
        self.epoch += 1                      # the next pass over the data draws a new (still rank-independent) file order
        for pin, pout in zip(inp_files, out_files):
            din = np.load(pin)
            dout = din if pout == pin else np.load(pout)
            a0, b0 = din[self.variables[0]], dout[self.out_variables[0]]
            for si, so in tile_slices(a0.shape[2], a0.shape[3], b0.shape[2], b0.shape[3], self.div, self.overlap):
                yield ({k: np.squeeze(din[k][:, :, si[0]:si[1], si[2]:si[3]], axis=1) for k in self.variables},
                       {k: np.squeeze(dout[k][:, :, so[0]:so[1], so[2]:so[3]], axis=1) for k in self.out_variables},
                       self.variables, self.out_variables)


class Downscale(IterableDataset):
    def __init__(self, dataset):
        super().__init__()
        self.dataset = dataset

    def __iter__(self):
        for inp, out, variables, out_variables in self.dataset:
            yield ({k: torch.from_numpy(v.astype(np.float32)) for k, v in inp.items()},
                   {k: torch.from_numpy(v.astype(np.float32)) for k, v in out.items()}, variables, out_variables)


class Normalize:
    """per-variable (x - mean) / std (torchvision.transforms.Normalize on a 1-channel image)."""

    def __init__(self, mean, std):
        self.mean, self.std = float(np.asarray(mean).reshape(-1)[0]), float(np.asarray(std).reshape(-1)[0])

    def __call__(self, t):
        return (t - self.mean) / self.std


class LogTransform:
    """precipitation: metres -> mm, values <= 0.25 mm/day -> 0, log1p (reference data/precipmodule.py:4-42)."""

    def __init__(self, m2mm=True, LOG1P=True, thres_mm_per_day=0.25):
        self.m2mm, self.LOG1P, self.thres = m2mm, LOG1P, thres_mm_per_day
        self.mean, self.std = 0.0, 1.0

    def __call__(self, t):
        if self.m2mm:
            t = t * 1000.0
            t = torch.where(t <= self.thres, torch.zeros((), dtype=t.dtype), t)
        else:
            t = torch.where(t <= self.thres / 1000.0, torch.zeros((), dtype=t.dtype), t)
        return torch.log1p(t) if self.LOG1P else torch.log(t + torch.finfo(torch.float64).eps)


class IndividualDataIter(IterableDataset):
    def __init__(self, dataset, transforms, output_transforms, subsample=6):
        super().__init__()
        self.dataset, self.transforms, self.output_transforms, self.subsample = dataset, transforms, output_transforms, subsample

    def __iter__(self):
        for inp, out, variables, out_variables in self.dataset:
            n = {v.shape[0] for v in inp.values()} | {v.shape[0] for v in out.values()}
            assert len(n) == 1
            for i in range(0, n.pop(), self.subsample):
                x = {k: inp[k][i] for k in inp}
                y = {k: out[k][i] for k in out}
                if self.transforms is not None:
                    x = {k: self.transforms[k](x[k]) for k in x}
                if self.output_transforms is not None:
                    y = {k: self.output_transforms[k](y[k]) for k in y}
                yield x, y, variables, out_variables


class ShuffleIterableDataset(IterableDataset):
    """shuffle buffer with a private random stream keyed by (seed, epoch, data-parallel rank, worker): data-parallel ranks
    draw different streams, the ranks of one tensor-parallel group (same data-parallel rank) identical ones -- their
    partial products are summed as if the inputs were the same batch, so they must be"""

    def __init__(self, dataset, buffer_size, seed: int = 0, dp_rank=None):
        super().__init__()
        assert buffer_size > 0
        self.dataset, self.buffer_size = dataset, buffer_size
        self.seed, self.epoch, self._dp_rank = int(seed), 0, dp_rank

    def _rank(self) -> int:
        if callable(self._dp_rank):
            return int(self._dp_rank())
        return int(self._dp_rank or 0)

    def __iter__(self):
        wi = torch.utils.data.get_worker_info()
        wid = wi.id if wi is not None else 0
        rng = random.Random(((self.seed * 1000003 + self.epoch) * 8191 + self._rank()) * 131 + wid)
        self.epoch += 1
        buf = []
        for x in self.dataset:
            if len(buf) == self.buffer_size:
                i = rng.randint(0, self.buffer_size - 1)
                yield buf[i]
                buf[i] = x
            else:
                buf.append(x)
        rng.shuffle(buf)
        while buf:
            yield buf.pop()


def collate_fn(batch):
    """(x[B,V,h,w], y[B,C,H,W], in_variables, out_variables) (itermodule.py:451-469)."""
    def stack(d: Dict[str, torch.Tensor]):
        return torch.stack(tuple(d.values()))
    x = torch.stack([stack(b[0]) for b in batch])
    y = torch.stack([stack(b[1]) for b in batch])
    return x, y, list(batch[0][0].keys()), list(batch[0][1].keys())
