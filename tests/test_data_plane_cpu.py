"""npz data plane against golden vectors produced by the reference's own NpyReader
(tests/golden/tiling.npz, generator: tests/golden/make_golden.py) + unit checks of sharding / transforms."""
import os

import numpy as np
import pytest
import torch

from climate_learn.data import iterdataset as D


def _write_files(td):
    fi, fo = [], []
    for f in range(3):
        yy, xx = np.meshgrid(np.arange(16), np.arange(32), indexing="ij")
        lo = (f * 1e6 + yy * 1000 + xx).astype(np.float64)[None, None].repeat(2, 0)
        YY, XX = np.meshgrid(np.arange(64), np.arange(128), indexing="ij")
        hi = (f * 1e6 + YY * 1000 + XX).astype(np.float64)[None, None].repeat(2, 0)
        pi, po = os.path.join(td, "in_%d.npz" % f), os.path.join(td, "out_%d.npz" % f)
        np.savez(pi, a=lo, b=lo + 0.5)
        np.savez(po, c=hi)
        fi.append(pi)
        fo.append(po)
    return fi, fo


@pytest.mark.parametrize("div,ov", [(1, 0), (2, 2), (4, 3), (2, 1)])
def test_tiles_match_reference_reader(golden_dir, tmp_path, div, ov):
    z = np.load(os.path.join(golden_dir, "tiling.npz"))["tiles_div%d_ov%d" % (div, ov)]
    fi, fo = _write_files(str(tmp_path))
    rd = D.NpyReader(fi, fo, ["a", "b"], ["c"], data_par_size=1, div=div, overlap=ov, rank=0)
    rec = []
    for xin, yout, iv, ovars in rd:
        a, c = xin["a"], yout["c"]
        assert np.array_equal(xin["b"], a + 0.5) and iv == ["a", "b"] and ovars == ["c"]
        rec.append([a.shape[1], a.shape[2], a[0, 0, 0], a[0, -1, -1], c.shape[1], c.shape[2], c[0, 0, 0], c[0, -1, -1]])
    assert np.array_equal(np.array(rec, dtype=np.float64), z)
    (h, w), (H, W) = D.tile_dims(16, 32, 64, 128, div, ov)
    assert all(r[0] == h and r[1] == w and r[4] == H and r[5] == W for r in rec)


def test_sharding_covers_files_once_and_wraps():
    got = []
    for rank in range(2):
        for wid in range(2):
            mult, rem, a, b = D.shard_range(8, rank, 2, 2, wid)
            assert (mult, rem) == (1, 0)
            got += list(range(a, b))
    assert sorted(got) == list(range(8))
    # fewer files than shards: list is wrapped around (iterdataset.py:63-69)
    mult, rem, a, b = D.shard_range(3, 3, 4, 2, 1)
    assert (mult, rem) == (2, 2) and (a, b) == (7, 8)


def test_transforms_and_collate():
    t = torch.tensor([[0.0001, 0.0005], [0.002, 0.0]])           # metres/day
    lt = D.LogTransform()
    ref = torch.log1p(torch.where(t * 1000 <= 0.25, torch.zeros(()), t * 1000))
    assert torch.allclose(lt(t), ref)
    n = D.Normalize(np.array([2.0]), np.array([4.0]))
    assert torch.allclose(n(torch.tensor([6.0])), torch.tensor([1.0]))
    batch = [({"a": torch.ones(4, 8), "b": torch.zeros(4, 8)}, {"c": torch.ones(16, 32)}, ["a", "b"], ["c"])] * 3
    x, y, iv, ov = D.collate_fn(batch)
    assert x.shape == (3, 2, 4, 8) and y.shape == (3, 1, 16, 32) and iv == ["a", "b"] and ov == ["c"]


def test_pipeline_end_to_end(tmp_path):
    fi, fo = _write_files(str(tmp_path))
    tf = {"a": D.Normalize(0.0, 1000.0), "b": D.Normalize(0.5, 1000.0)}
    ds = D.ShuffleIterableDataset(
        D.IndividualDataIter(D.Downscale(D.NpyReader(fi, fo, ["a", "b"], ["c"], div=2, overlap=2, rank=0)), tf,
                             {"c": D.Normalize(0.0, 1.0)}, subsample=1), buffer_size=5)
    items = list(ds)
    assert len(items) == 3 * 4 * 2            # files x tiles x timesteps
    x, y, iv, ov = D.collate_fn(items[:4])
    (h, w), (H, W) = D.tile_dims(16, 32, 64, 128, 2, 2)
    assert x.shape == (4, 2, h, w) and y.shape == (4, 1, H, W) and x.dtype == torch.float32
    assert torch.allclose(x[:, 0], x[:, 1])   # both normalise to the same field


def test_iterdatamodule_on_disk(tmp_path):
    """IterDataModule over a reference-format directory tree: dims, lat/lon, climatology, batches."""
    import climate_learn as cl
    lo_root, hi_root = tmp_path / "lo", tmp_path / "hi"
    rng = np.random.default_rng(0)
    iv = ["land_sea_mask", "orography", "lattitude", "landcover", "total_precipitation_24hr"]
    ov = ["total_precipitation_24hr"]
    for root, (H, W), vs in ((lo_root, (16, 32), iv), (hi_root, (64, 128), ov)):
        for split in ("train", "val"):
            os.makedirs(root / split)
            for sh in range(2):
                np.savez(root / split / ("2000_%d.npz" % sh), **{v: np.abs(rng.normal(size=(3, 1, H, W))) * 1e-3 for v in vs})
            np.savez(root / split / "climatology.npz", **{v: np.zeros((1, H, W)) for v in vs})
        np.save(root / "lat.npy", np.linspace(-80, 80, H))
        np.save(root / "lon.npy", np.linspace(0, 350, W))
        np.savez(root / "normalize_mean.npz", **{v: np.array([0.5e-3]) for v in vs})
        np.savez(root / "normalize_std.npz", **{v: np.array([1e-3]) for v in vs})
    dm = cl.data.IterDataModule("downscaling", str(lo_root), str(hi_root), iv, ov, batch_size=4, buffer_size=8,
                                div=2, overlap=2, subsample=1)
    assert dm.get_lat_lon() == (None, None)
    dm.setup()
    (h, w), (H, W) = D.tile_dims(16, 32, 64, 128, 2, 2)
    din, dout = dm.get_data_dims()
    assert tuple(din) == (4, 5, h, w) and tuple(dout) == (4, 1, H, W)
    assert len(dm.get_lat_lon()[0]) == 64 and dm.get_climatology("train")[ov[0]].shape == (64, 128)
    batches = list(dm.train_dataloader())
    assert sum(b[0].shape[0] for b in batches) == 2 * 4 * 3        # shards x tiles x timesteps
    x, y, a, b = batches[0]
    assert x.shape[1:] == (5, h, w) and y.shape[1:] == (1, H, W) and a == iv and b == ov
    assert float(y.min()) >= 0.0                                    # log1p precipitation
    # a loader that is RE-ITERATED draws a new order every pass, also with worker processes (they iterate copies of the
    # datasets: the loader itself numbers its passes, round-2 advisor), and set_epoch() reproduces a pass
    dmw = cl.data.IterDataModule("downscaling", str(lo_root), str(hi_root), iv, ov, batch_size=4, buffer_size=8,
                                 div=2, overlap=2, subsample=1, num_workers=1)
    dmw.setup()
    ld = dmw.train_dataloader()
    key = lambda bs: [float(v) for b_ in bs for v in b_[1].reshape(b_[1].shape[0], -1).sum(1)]
    p0, p1 = key(list(ld)), key(list(ld))
    assert sorted(p0) == sorted(p1) and p0 != p1                     # the same samples, another order
    ld.set_epoch(0)
    assert key(list(ld)) == p0


def _shard_worker(rank, world, root, port, q):
    """dp2 x tp2 layout of the reference driver (tensor-parallel ranks adjacent): every rank reads its on-disk shard"""
    import torch.distributed as dist
    import climate_learn as cl
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tp = 2
    dp_group = None
    for i in range(tp):
        g = dist.new_group([i + j * tp for j in range(world // tp)])
        dp_group = g if rank % tp == i else dp_group
    iv = ["land_sea_mask", "orography", "lattitude", "landcover", "total_precipitation_24hr"]
    dm = cl.data.IterDataModule("downscaling", os.path.join(root, "lo"), os.path.join(root, "hi"), iv,
                                ["total_precipitation_24hr"], data_par_size=world // tp, data_par_group=dp_group,
                                batch_size=2, buffer_size=3, subsample=1, seed=5)
    dm.setup()
    epochs = []
    for _ in range(2):
        ids = []
        for x, y, _, _ in dm.train_dataloader():
            ids += [float(v) for v in x[:, 1, 0, 0]]          # the orography field carries the sample id
        epochs.append(ids)
    q.put((rank, epochs))
    dist.barrier()
    dist.destroy_process_group()


def test_two_dp_ranks_read_disjoint_shards_and_tp_peers_identical_batches(tmp_path):
    """the file order must be the same permutation on every rank (disjoint, complete data-parallel
    shards) and the shuffle-buffer stream must depend on the DATA-parallel rank only (the ranks of a tensor-parallel group
    sum partial products of what must be the same batch)."""
    import socket
    import torch.multiprocessing as mp
    rng = np.random.default_rng(0)
    iv = ["land_sea_mask", "orography", "lattitude", "landcover", "total_precipitation_24hr"]
    nfiles, T = 6, 2
    for root, (H, W), vs in ((tmp_path / "lo", (8, 16), iv), (tmp_path / "hi", (32, 64), ["total_precipitation_24hr"])):
        os.makedirs(root / "train")
        for sh in range(nfiles):
            d = {v: np.abs(rng.normal(size=(T, 1, H, W))) * 1e-3 for v in vs}
            if "orography" in d:
                for t in range(T):
                    d["orography"][t] = sh * 100 + t               # sample id, survives Normalize(0, 1)
            np.savez(root / "train" / ("2000_%d.npz" % sh), **d)
        np.save(root / "lat.npy", np.linspace(-80, 80, H))
        np.save(root / "lon.npy", np.linspace(0, 350, W))
        np.savez(root / "normalize_mean.npz", **{v: np.array([0.0]) for v in vs})
        np.savez(root / "normalize_std.npz", **{v: np.array([1.0]) for v in vs})
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 4, str(tmp_path), port, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(4))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    every = sorted(sh * 100 + t for sh in range(nfiles) for t in range(T))
    for ep in range(2):
        assert got[0][ep] == got[1][ep] and got[2][ep] == got[3][ep]       # tensor-parallel peers: the same batches
        a, b = got[0][ep], got[2][ep]                                       # the two data-parallel ranks
        assert not set(a) & set(b) and sorted(a + b) == every               # disjoint and complete
    assert got[0][0] != got[0][1]                                           # a new epoch reshuffles
