"""Host logic of the head-split tensor parallelism (climate_learn.dist.tp): state_dict shard / merge,
the rank layout of the driver's init_par_groups, and the two autograd collectives over a world-2 gloo group."""
import importlib.util
import os
import socket
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(tp=1, grp=None, heads=4, D=128):
    from climate_learn.models.hub import Res_Slim_ViT
    consts = ["land_sea_mask", "orography", "lattitude", "landcover"]
    return Res_Slim_ViT(consts + ["total_precipitation_24hr"], (16, 32), 5, 1, 1, patch_size=2, embed_dim=D, depth=2,
                        decoder_depth=1, num_heads=heads, tensor_par_size=tp, tensor_par_group=grp)


def test_shard_merge_roundtrip_and_reference_shapes():
    from climate_learn.dist import tp
    torch.manual_seed(0)
    full = {k: torch.randn_like(v) for k, v in _model().state_dict().items()}
    D, H = 128, 4
    for n in (2, 4):
        shards = [tp.shard_state_dict(full, n, r, H) for r in range(n)]
        s0 = shards[0]
        # the shapes the reference builds with tensor_par_size = n (attention.py:36-40,118-127, mlp.py:50-55)
        assert s0["blocks.0.attn.qkv.weight"].shape == (3 * D // n, D)
        assert s0["blocks.0.attn.qkv.bias"].shape == (3 * D // n,)
        assert s0["blocks.0.attn.proj.weight"].shape == (D, D // n)
        assert s0["blocks.0.mlp.fc1.weight"].shape == (4 * D // n, D)
        assert s0["blocks.0.mlp.fc2.weight"].shape == (D, 4 * D // n)
        assert s0["var_agg.q.weight"].shape == (D // n, D)
        assert s0["var_agg.kv.weight"].shape == (2 * D // n, D)
        assert s0["var_agg.proj.weight"].shape == (D, D // n)
        assert s0["blocks.0.attn.proj.bias"].shape == (D,) and s0["head.0.weight"].shape == (D, D)
        # head h of the full qkv lands on rank h // (H/n): q rows of rank 1 = q rows of its first head
        d = D // H
        hl = H // n
        assert torch.equal(shards[1]["blocks.1.attn.qkv.weight"][:d], full["blocks.1.attn.qkv.weight"][hl * d:(hl + 1) * d])
        assert torch.equal(shards[1]["blocks.1.attn.qkv.weight"][hl * d:(hl + 1) * d],
                           full["blocks.1.attn.qkv.weight"][D + hl * d:D + (hl + 1) * d])       # its k rows
        # the row-parallel biases sum to the original
        assert torch.equal(sum(s["blocks.0.mlp.fc2.bias"] for s in shards), full["blocks.0.mlp.fc2.bias"])
        merged = tp.merge_state_dicts(shards, H)
        assert set(merged) == set(full)
        for k in full:
            assert torch.equal(merged[k], full[k]), k
    with pytest.raises(ValueError):
        tp.shard_state_dict(full, 3, 0, H)


def test_tp_needs_a_group_of_that_size():
    with pytest.raises(ValueError):
        _model(tp=2, grp=None)


def _load_driver():
    spec = importlib.util.spec_from_file_location("o2_driver", os.path.join(ROOT, "examples", "intermediate_downscaling.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _worker(rank, world, port, q):
    try:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        from climate_learn.dist import tp
        drv = _load_driver()
        # world 2 = one tensor-parallel group of 2, data-parallel degree 1 (reference layout :173-183,:212-247)
        dpg, tpg = drv.init_par_groups(rank, 1, 2, 1, 1, 1, world, num_heads=4)
        assert dist.get_world_size(tpg) == 2 and dist.get_rank(tpg) == rank
        assert dist.get_world_size(dpg) == 1 and dist.get_rank(dpg) == 0
        # and the pure data-parallel layout
        dpg2, tpg2 = drv.init_par_groups(rank, 2, 1, 1, 1, 2, world, num_heads=4)
        assert tpg2 is None and dist.get_world_size(dpg2) == 2
        # model construction with the group: local shapes, tags
        m = _model(tp=2, grp=tpg)
        assert m.blocks[0].attn.qkv.weight.shape == (3 * 128 // 2, 128)
        assert m.blocks[0].attn.qkv.weight._o2_tp == "qkv" and not hasattr(m.norm.weight, "_o2_tp")
        # replicated-parameter sync: start different, end equal; split tensors are left alone
        with torch.no_grad():
            for p in m.parameters():
                p.add_(float(rank + 1))
        before = m.blocks[0].mlp.fc1.weight.clone()
        tp.sync_replicated(m, tpg)
        got = [None, None]
        dist.all_gather_object(got, {k: v.clone() for k, v in m.state_dict().items()}, group=tpg)
        for k in got[0]:
            if tp.split_kind(k) is None and not k.endswith(tp._SUMMED_BIASES):
                assert torch.equal(got[0][k], got[1][k]), k
        assert torch.equal(m.blocks[0].mlp.fc1.weight, before)
        # the two autograd collectives
        x = torch.full((3, 4), float(rank + 1), requires_grad=True)
        y = tp.AllReduceFwdIdentityBwd.apply(tp.IdentityFwdAllReduceBwd.apply(x, tpg) * (rank + 1.0), tpg)
        assert torch.equal(y, torch.full((3, 4), 1.0 * 1 + 2.0 * 2))            # sum_r x_r * (r+1)
        (y * (10.0 if rank == 0 else 100.0)).sum().backward()
        # backward: identity through the output, x(r+1), then SUM over ranks: 10*1 + 100*2
        assert torch.equal(x.grad, torch.full((3, 4), 210.0)), x.grad
        t = torch.tensor([float(rank)])
        assert float(tp.all_reduce_max(t, None)) == 1.0
        # ReplicaGuard: what keeps tensor-parallel replicas identical (advisor r4: do not rely on bitwise agreement where the
        # reduction order is not pinned)
        from climate_learn import _hip
        os.environ.pop("ORBIT2_TP_REPLICA_SYNC", None)
        same = [torch.arange(40, dtype=torch.float32).to(torch.bfloat16), torch.linspace(-1, 1, 17)]
        g = tp.ReplicaGuard(tpg, data_world=1, check_every=3)             # the group is the whole job, fixed-order kernels
        assert not g.needs_broadcast()
        for _ in range(4):
            g.after_reduction([v.clone() for v in same])
        assert g.checks == 2 and g.broadcasts == 0                        # first step and every third
        bad = [v.clone() for v in same]
        if rank == 1:
            bad[1][5] += 1e-7 * (1 + bad[1][5].abs())                     # one ulp-scale difference on one rank
        g2 = tp.ReplicaGuard(tpg, data_world=1)
        try:
            g2.after_reduction(bad)
            raise AssertionError("a replica mismatch went unnoticed")
        except RuntimeError as e:
            assert "replicas disagree" in str(e) and "[1]" in str(e)
        # two words that differ by OPPOSITE raw amounts (and a permutation) cancel in a plain sum of the words: the
        # position-weighted checksum must still see them (advisor, round 5)
        for kind in ("opposite", "swap"):
            canc = [v.clone() for v in same]
            if rank == 1:
                w = canc[0].view(torch.int16)
                if kind == "opposite":
                    w[3] += 1; w[7] -= 1
                else:
                    w[[3, 7]] = w[[7, 3]]
            try:
                tp.ReplicaGuard(tpg, data_world=1).after_reduction(canc)
                raise AssertionError("a replica mismatch that cancels in the plain sum went unnoticed (%s)" % kind)
            except RuntimeError as e:
                assert "replicas disagree" in str(e) and "[0]" in str(e)
        g3 = tp.ReplicaGuard(tpg, data_world=2)                           # data parallelism beside the group: exchange
        assert g3.needs_broadcast()
        g3.after_reduction(bad)
        assert g3.broadcasts == 1 and torch.equal(bad[1], same[1])       # every rank now holds the first rank's range
        got = [None, None]
        dist.all_gather_object(got, bad[1], group=tpg)
        assert torch.equal(got[0], got[1])
        _hip.atomics_in_grad_path = True                                  # an atomics-accumulating kernel was used: exchange too
        assert tp.ReplicaGuard(tpg, data_world=1).needs_broadcast()
        _hip.atomics_in_grad_path = False
        os.environ["ORBIT2_TP_REPLICA_SYNC"] = "off"
        g4 = tp.ReplicaGuard(tpg, data_world=2)
        g4.after_reduction(bad)
        assert g4.broadcasts == g4.checks == 0
        os.environ.pop("ORBIT2_TP_REPLICA_SYNC")
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_groups_sync_and_autograd_collectives_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(30)
    for r in res:
        assert r[1] == "ok", r[1]
