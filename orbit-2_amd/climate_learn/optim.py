"""Fused AdamW + loss scaler for the HIP path.

HipAdamW has torch.optim.AdamW's math (utils/loaders.py:398-399 of the reference selects torch AdamW) and the
torch.optim.Optimizer surface (param_groups / state_dict / zero_grad) so LR schedulers drive it unchanged.
With a HipDataParallel-managed model one fused kernel per flat range updates fp32 master, both moments and the
bf16 compute copy, and folds in the gradient averaging (1/world) and loss-scale division.
HipGradScaler mirrors ShardedGradScaler(init_scale=8192, growth_interval=100) + the script's min-scale floor
(examples/intermediate_downscaling.py:493-497,732-742)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from .dist import tp as _tp

from . import _hip

F32 = torch.float32


class HipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, engine=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.engine = engine
        self._step = 0
        self.found_inf = None
        self.grad_scale = 1.0
        if engine is not None:
            # moments cover what this rank updates: everything (flat32 layout), or with a sharded optimizer its chunk
            # of every unit's bf16 range plus the replicated fp32-compute parameters
            self.m = torch.zeros(engine.opt_state_size, dtype=F32, device=engine.device)
            self.v = torch.zeros(engine.opt_state_size, dtype=F32, device=engine.device)
            self.found_inf = torch.zeros(1, dtype=F32, device=engine.device)

    @torch.no_grad()
    def step(self, closure=None):
        self._step += 1
        g = self.param_groups[0]
        lr, (b1, b2), eps, wd = g["lr"], g["betas"], g["eps"], g["weight_decay"]
        if self.engine is not None:
            e = self.engine
            e.finish_grad_sync()
            gs = self.grad_scale / e.grad_world
            fi = self.found_inf if self.check_inf else None
            for sg in e.opt_segments:
                n, os_ = sg["n"], sg["os"]
                _hip.adamw(sg["p32"], self.m[os_:], self.v[os_:], sg["g"], sg["p16"], n, lr, b1, b2, eps, wd, self._step,
                           gs, fi)
            e.gather_params()
            return None
        # un-managed parameters (unit tests / tiny models): one launch per tensor
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["m"] = torch.zeros_like(p, dtype=F32)
                    st["v"] = torch.zeros_like(p, dtype=F32)
                gr = p.grad.contiguous()
                p16 = getattr(p, "_o2c", None)
                _hip.adamw(p.data, st["m"], st["v"], gr, p16, p.numel(), group["lr"], group["betas"][0],
                           group["betas"][1], group["eps"], group["weight_decay"], self._step, self.grad_scale, None)
        return None

    check_inf = False

    def zero_grad(self, set_to_none: bool = True):
        if self.engine is not None:
            self.engine.zero_grad()
        else:
            super().zero_grad(set_to_none)

    # ---- checkpoint format: torch.optim.AdamW's ------------------------------------------------------------------
    # state[i] = {step, exp_avg, exp_avg_sq} per parameter, i = position in param_groups (= model.parameters() order, the
    # reference's).  It does not depend on the engine's flat layout, so a checkpoint written by the replicated engine, by
    # the sharded one at any world size (whose unit ranges are padded to world * 128 elements), or by the reference's
    # torch.optim.AdamW (`optimizer_state_dict` of its .ckpt files) loads into any of them.
    def _params(self):
        return [p for g in self.param_groups for p in g["params"]]

    def _moments_per_param(self, st, to_cpu: bool = False):
        """{id(param): tensor of the parameter's shape} from a moment buffer laid out like the engine's optimizer segments
        (a segment that is a rank's 1/N chunk of a range is all-gathered first: a collective in the sharded modes).
        to_cpu: every gathered range goes to the host before the next one is gathered (checkpoints of sharded engines)"""
        out = {}
        for sg in self.engine.opt_segments:
            rng = st[sg["os"]:sg["os"] + sg["n"]]
            if sg["gather"]:
                rng = self.engine.gather_range(rng)
            for p, off, k in sg["members"]:
                v = rng[off:off + k].view(p.shape)
                out[id(p)] = v.cpu() if to_cpu else v
            del rng
        return out

    def _load_moments_per_param(self, st, per_param):
        """the inverse: per_param maps id(param) -> full tensor (or is missing: zeros)"""
        e = self.engine
        for sg in e.opt_segments:
            n = sg["n"]
            full = torch.zeros(n * (e.world if sg["gather"] else 1), dtype=F32, device=st.device)
            for p, off, k in sg["members"]:
                t = per_param.get(id(p))
                if t is not None:
                    full[off:off + k].copy_(t.reshape(-1).to(st.device, F32))
            st[sg["os"]:sg["os"] + n].copy_(full[e.rank * n:(e.rank + 1) * n] if sg["gather"] else full)

    def state_dict(self, offload_to_cpu: bool = False):
        if self.engine is None:
            sd = super().state_dict()
            for st in sd["state"].values():                     # torch key names for the un-managed path too
                if "m" in st:
                    st["exp_avg"], st["exp_avg_sq"] = st.pop("m"), st.pop("v")
                    st["step"] = torch.tensor(float(self._step))
            sd["orbit2"] = {"step": self._step, "format": 2}
            return sd
        pm, pv = self._moments_per_param(self.m, offload_to_cpu), self._moments_per_param(self.v, offload_to_cpu)
        state, idx = {}, []
        for i, p in enumerate(self._params()):
            state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": pm[id(p)], "exp_avg_sq": pv[id(p)]}
            idx.append(i)
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        groups[0]["params"] = idx
        return {"state": state, "param_groups": groups, "orbit2": {"step": self._step, "format": 2}}

    def load_state_dict(self, sd):
        sd = dict(sd)                                    # the caller's dict is left as it was
        extra = sd.pop("orbit2", None)
        if self.engine is None:
            sd["state"] = {k: ({"m": v["exp_avg"], "v": v["exp_avg_sq"]} if "exp_avg" in v else dict(v))
                           for k, v in sd.get("state", {}).items()}
            super().load_state_dict(sd)
            steps = [float(v["step"]) for v in sd.get("state", {}).values() if "step" in v]
            self._step = extra["step"] if extra else (int(max(steps)) if steps else 0)
            return
        for g, sg in zip(self.param_groups, sd.get("param_groups", [])):       # hyper-parameters (lr schedule position)
            g.update({k: v for k, v in sg.items() if k != "params"})
        if extra and "m" in extra:
            raise RuntimeError("this checkpoint holds flat optimizer moments in its writer's buffer layout (round-1 format): "
                               "not loadable; re-save it with the current code")
        state = sd.get("state", {})
        params = self._params()
        if not state:
            import warnings
            warnings.warn("optimizer checkpoint holds no per-parameter state: AdamW moments and step count start from zero")
            self._step = extra["step"] if extra else 0
            return
        pm, pv, steps = {}, {}, []
        for i, p in enumerate(params):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise RuntimeError("optimizer state %d has shape %s, parameter has %s"
                                   % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
            pm[id(p)], pv[id(p)] = st["exp_avg"], st["exp_avg_sq"]
            steps.append(float(st["step"]))
        self._load_moments_per_param(self.m, pm)
        self._load_moments_per_param(self.v, pv)
        self._step = int(extra["step"]) if extra else (int(max(steps)) if steps else 0)


class HipGradScaler:
    """Dynamic loss scaling: scale(loss) -> backward -> step(optimizer) -> update()."""

    def __init__(self, init_scale=8192.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=100,
                 min_scale=128.0, process_group=None, sync_world=False):
        self._scale = float(init_scale)
        self.growth_factor, self.backoff_factor = growth_factor, backoff_factor
        self.growth_interval, self.min_scale = growth_interval, min_scale
        self._good = 0
        self.pg = process_group
        self.sync_world = sync_world     # found_inf is agreed over ALL ranks (tensor-parallel shards differ per rank)
        self._found = None
        self._opt = None                 # the optimizer of the last step() (update() rolls its step count back on overflow)

    def get_scale(self):
        return self._scale

    def scale(self, loss):
        return loss * self._scale

    def step(self, optimizer: HipAdamW):
        eng = optimizer.engine
        assert eng is not None, "HipGradScaler needs a HipDataParallel-managed optimizer"
        eng.finish_grad_sync()
        fi = optimizer.found_inf
        fi.zero_()
        for sg in eng.opt_segments:       # with a sharded optimizer: this rank's reduced chunk of every bf16 bucket
            _hip.check_finite(sg["g"], sg["n"], fi)
        if self.sync_world and dist.is_initialized() and dist.get_world_size() > 1:
            _tp.all_reduce_max(fi, None)
        elif getattr(eng, "grad_world", eng.world) > 1:
            for grp in getattr(eng, "found_inf_groups", [eng.pg]):     # shard group, then (HYBRID) the replica group
                dist.all_reduce(fi, op=dist.ReduceOp.MAX, group=grp)
        optimizer.grad_scale = 1.0 / self._scale
        optimizer.check_inf = True
        optimizer.step()            # the kernel skips the update on device when found_inf != 0
        optimizer.check_inf = False
        optimizer.grad_scale = 1.0
        self._found = fi
        self._opt = optimizer

    def update(self):
        bad = bool(self._found.item() != 0.0) if self._found is not None else False
        if bad:
            # the kernel skipped the update on device: like torch's GradScaler (which does not call optimizer.step() on an
            # overflow) the step count behind the bias corrections must not advance either
            if self._opt is not None:
                self._opt._step -= 1
            self._scale = max(self._scale * self.backoff_factor, self.min_scale)
            self._good = 0
        else:
            self._good += 1
            if self._good >= self.growth_interval:
                self._scale *= self.growth_factor
                self._good = 0
        self._opt = None
        return bad
