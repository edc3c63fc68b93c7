"""2-process (one card, gloo) stage-by-stage comparison of the tensor-parallel model against the single-rank one."""
import os, sys, traceback
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "orbit-2_amd"), os.path.join(ROOT, "tests")]

def nerr(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))

def worker(rank, world, port):
    try:
        from test_tp_gpu import _build, GOLDEN
        from test_model_gpu import CASES
        from climate_learn.dist import tp
        from climate_learn import _ops
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        torch.cuda.set_device(0)
        grp = dist.new_group(list(range(world)))
        tag = "v5c1_hd64"; c = CASES[tag]
        z = np.load(os.path.join(GOLDEN, "model_%s.npz" % tag))
        full = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
        m = _build(c, world, grp); m.load_state_dict(tp.shard_state_dict(full, world, rank, c["heads"])); m = m.cuda().eval()
        r = _build(c, 1, None); r.load_state_dict(full); r = r.cuda().eval()
        x = torch.from_numpy(z["x"]).cuda().float().contiguous()
        with torch.no_grad():
            ids = m.get_var_ids(tuple(c["in_vars"]))
            st, gt = m._tables(ids); st1, gt1 = r._tables(ids)
            H, Hl = c["heads"], c["heads"] // world
            print(rank, "stab", nerr(st, st1[rank * Hl:(rank + 1) * Hl]), "gtab", nerr(gt, gt1[..., rank * gt.shape[-1]:(rank + 1) * gt.shape[-1]]), flush=True)
            t = _ops.EmbedFn.apply(x, st, gt, m._posres(), m.var_agg.proj.weight, m.var_agg.proj.bias, Hl, 0.0, grp)
            t1 = _ops.EmbedFn.apply(x, st1, gt1, r._posres(), r.var_agg.proj.weight, r.var_agg.proj.bias, H, 0.0)
            print(rank, "embed", nerr(t, t1), flush=True)
            for i, (b, b1) in enumerate(zip(m.blocks, r.blocks)):
                t, t1 = b(t1.clone()), b1(t1)
                print(rank, "block", i, nerr(t, t1), flush=True)
            print(rank, "norm of last", float(t.float().norm()), float(t1.float().norm()), flush=True)
            p = m(x, c["in_vars"], c["out_vars"]); p1 = r(x, c["in_vars"], c["out_vars"])
            print(rank, "pred", nerr(p, p1), "vs golden", nerr(p, torch.from_numpy(z["pred"])), nerr(p1, torch.from_numpy(z["pred"])), flush=True)
    except Exception:
        traceback.print_exc()
    finally:
        dist.destroy_process_group()

if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29611), nprocs=2)
