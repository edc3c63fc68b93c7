"""Parameter-sharding data-parallel engine: the MI355X counterpart of the reference's FSDP FULL_SHARD / HYBRID_SHARD wrap
(examples/intermediate_downscaling.py:609-617: `parallelism.fsdp > 1`, with `simple_ddp > 1` for HYBRID;
transformer_auto_wrap_policy({Block, Sequential}), bf16 MixedPrecision, forward_prefetch=True).

One process per GPU, as `HipDataParallel` (dp_engine.py, the NO_SHARD engine), same parameter plumbing for the kernels
(`param._o2c` bf16 compute copy, `param._o2g` bf16 gradient view, `grad_ready` notifications) -- but a rank keeps only
1/N of every parameter-sharded unit:

    chunk32   fp32 master of the rank's chunk of every sharded unit's bf16-compute range (+ both AdamW moments, in the optimizer)
    chunk16   its bf16 compute copy     -- the all-gather source
    gchunk16  the rank's chunk of the reduced gradient  -- the reduce-scatter destination

A unit (one per Block, one for the head Sequential) exists in full only while it is used: its chunks are ALL-GATHERED into a
buffer of a small pool right before its forward (`_ops.unit_enter`) and again right before its backward
(`_UnitBackwardGate`), on the communication stream and ONE UNIT AHEAD of the compute stream (the next unit in the recorded
execution order is prefetched while the current one computes; the reference sets forward_prefetch=True); the buffer goes
back to the pool when the unit's forward (or backward: a second gate on the unit's input) has returned.  The backward kernels write the unit's weight gradients into a
pooled gradient buffer; when the last one is in, the buffer is REDUCE-SCATTERED over the shard group into `gchunk16` (and, for
HYBRID_SHARD, the chunk is all-reduced over the replica group), overlapping the rest of backward.  AdamW then runs on the
chunks only.  The root unit (embeddings, final norm: used at both ends of the step) and the fp32-compute parameters
(convolutions, variable-aggregation tables; < 1 % of the model) stay resident and replicated, with all-reduced gradients.

Per-rank persistent bytes of a sharded range: (4 + 4 + 4 + 2 + 2) / N per parameter instead of 18 (master, two moments,
compute copy, gradient); `param_bytes_per_rank()` reports both.  `gloo` (CPU tests) has no reduce-scatter: the all-reduce
leaves the same sum in the rank's chunk.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from .dp_engine import BF, F32, Bucket, CommStats, _ALIGN, _round_up, default_units


class _Unit(Bucket):
    def __init__(self, name):
        super().__init__(name)
        self.module: Optional[nn.Module] = None
        self.sharded = False
        self.n = 0                 # length of the bf16-compute range (padded to world * 128)
        self.ck = 0                # chunk length
        self.cs = 0                # offset of the rank's chunk in the chunk buffers
        self.members: List[Tuple[nn.Parameter, int, int]] = []     # (param, offset in range, numel)
        self.pbuf = None           # index of the pooled parameter buffer holding the gathered unit (or None)
        self.pevent = None         # communication-stream event: gather done
        self.gbuf = None           # index of the pooled gradient buffer
        self.launched = False
        self.gaps: List[Tuple[int, int]] = []      # [start, end) ranges of the unit's range no backward kernel writes
        self.layout = ()           # signature of the member layout: units with equal signatures write the same ranges


class HipFullyShardedDataParallel(nn.Module):

    # communication accounting (CommStats) records timing-enabled events and reads them back: not inside a hipGraph capture
    @property
    def comm_stats(self):
        return self.__dict__.get("_comm_stats")

    @comm_stats.setter
    def comm_stats(self, v):
        if v is not None and self.__dict__.get("_o2_capture_live", False):
            raise RuntimeError("comm_stats cannot be switched on while a hipGraph capture of this engine's step exists "
                               "(GraphedTrainStep): captured events carry no timestamps")
        self.__dict__["_comm_stats"] = v

    # The communication stream.  In SINGLE-STREAM mode (`single_stream()`, switched on by GraphedTrainStep) the property reads
    # None and every `if self.comm_stream is not None` below takes the plain path: gathers, reduce-scatters and buffer hand-overs
    # are queued on the caller's stream in program order, no cross-stream event is created.  That is the form a hipGraph capture of
    # this engine's step takes: the two-stream form records ~4 fork / join edges PER UNIT in both directions between the capturing
    # stream and the communication stream (pool-release event -> gather, gather-done event -> compute, gradient-ready event ->
    # reduce-scatter, reduce-scatter-done event -> next user of the buffer), every one of them event-joined before the step ends
    # (list in DESIGN 5) -- and hipStreamEndCapture segfaulted on it in rounds 3 and 5.  A replayed graph of a launch-bound step
    # gains nothing from a second stream (its collectives are microseconds), so the capture does without.
    @property
    def comm_stream(self):
        return None if self.__dict__.get("_o2_single_stream", False) else self.__dict__.get("_comm_stream")

    @comm_stream.setter
    def comm_stream(self, v):
        self.__dict__["_comm_stream"] = v

    def single_stream(self, on: bool = True):
        """queue the engine's communication on the caller's stream from now on (see `comm_stream`); call between steps"""
        real = self.__dict__.get("_comm_stream")
        if real is not None:                            # nothing of an earlier two-stream step may still be in flight
            torch.cuda.current_stream().wait_stream(real)
            real.wait_stream(torch.cuda.current_stream())
        if any(u.pbuf is not None or u.gbuf is not None for u in self.sharded_units):
            raise RuntimeError("single_stream(): a unit still holds a pooled buffer -- call it between steps")
        self._pfree_ev = [None] * len(self._pfree_ev)
        self._gfree_ev = [None, None]
        for u in self.units:
            u.pevent = None
        self.__dict__["_o2_single_stream"] = bool(on)
    def __init__(self, module: nn.Module, process_group=None, unit_types: Tuple[type, ...] = (), is_lowp=None,
                 sync_module_states: bool = True, replicate_group=None, prefetch: bool = True, pool_size: int = 3,
                 tp_group=None):
        """process_group: the SHARD group (all data-parallel ranks for FULL_SHARD, the reference's fsdp_group for HYBRID);
        replicate_group: the group of ranks holding the same chunk (the reference's simple_ddp_group; HYBRID only);
        tp_group: this rank's tensor-parallel group (dist/tp.py; the reference's 2-D FSDP x TP layout, configs/interm_1b.yaml:
        14-24).  The model then holds the rank's head / column slices; the shard and replicate groups are the data-parallel
        ranks of ONE tensor-parallel column, so chunks are cut from the rank's own slices and the tensor-parallel collectives of
        the model are untouched.  Parameters that are NOT split over tp_group (LayerNorms, embeddings, convolutions) are
        replicas: laid out first inside every unit; whether their reduced gradient ranges are exchanged over the group or only
        verified is decided by dist/tp.py ReplicaGuard (see HipDataParallel)."""
        super().__init__()
        self.module = module
        self.pg = process_group
        self.tp_group = tp_group if (tp_group is not None and dist.is_initialized()
                                     and dist.get_world_size(tp_group) > 1) else None
        self.rg = replicate_group if (replicate_group is not None and dist.get_world_size(replicate_group) > 1) else None
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.replicas = dist.get_world_size(self.rg) if self.rg is not None else 1
        self.grad_world = self.world * self.replicas
        self.found_inf_groups = [self.pg] + ([self.rg] if self.rg is not None else [])
        from . import tp as _tp
        self.replica_guard = _tp.ReplicaGuard(self.tp_group, self.grad_world) if self.tp_group is not None else None
        self._replicas_synced = False
        self.shard = True
        self.shard_params = True
        import os as _os
        self.force_comm = dist.is_initialized() and _os.environ.get("ORBIT2_FORCE_COLLECTIVES", "0") == "1"
        self.comm = self.world > 1 or self.force_comm
        self.prefetch = prefetch
        params = list(module.parameters())
        assert params, "module has no parameters"
        dev = params[0].device
        self.device = dev
        if is_lowp is None:
            is_lowp = lambda name, p: getattr(p, "_o2_lowp", False)
        N = self.world
        # every rank starts from rank 0's weights (FSDP sync_module_states=True)
        if sync_module_states and self.comm:
            src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0
            for p in params:
                dist.broadcast(p.data, src=src, group=self.pg)
            if self.rg is not None:
                srcr = dist.get_global_rank(self.rg, 0)
                for p in params:
                    dist.broadcast(p.data, src=srcr, group=self.rg)
        # ---- layout
        self.units: List[_Unit] = []
        self.buckets = self.units                      # name used by the shared tests / tools
        self._unit_of: Dict[int, _Unit] = {}
        n_chunk = n_res16 = n_res32 = n_g32 = 0
        plan = []
        for uname, ps in default_units(module, unit_types):
            u = _Unit(uname)
            lo = [(n, p) for n, p in ps if is_lowp(n, p)]
            hi = [(n, p) for n, p in ps if not is_lowp(n, p)]
            if self.tp_group is not None:           # replicas first, tensor-parallel slices after (stable order)
                lo.sort(key=lambda np_: hasattr(np_[1], "_o2_tp"))
                hi.sort(key=lambda np_: hasattr(np_[1], "_o2_tp"))
            u.sharded = bool(lo) and uname != "root"
            off = 0
            u.rep_lo = 0                            # length of the replicated prefix of the unit's bf16 range
            for n, p in lo:
                u.members.append((p, off, p.numel()))
                off += _round_up(p.numel())
                if not hasattr(p, "_o2_tp"):
                    u.rep_lo = off
            u.n = _round_up(off, N * _ALIGN) if lo else 0
            u.ck = u.n // N
            if u.sharded:
                u.cs, n_chunk = n_chunk, n_chunk + u.ck
                u.module = module.get_submodule(uname)
            else:
                n_res16 += u.n
            n_res32 += (0 if u.sharded else u.n) + sum(_round_up(p.numel()) for _, p in hi)
            n_g32 += sum(_round_up(p.numel()) for _, p in hi)
            plan.append((u, lo, hi))
        self.chunk32 = torch.zeros(max(n_chunk, 1), dtype=F32, device=dev)
        self.chunk16 = torch.zeros(max(n_chunk, 1), dtype=BF, device=dev)
        self.gchunk16 = torch.zeros(max(n_chunk, 1), dtype=BF, device=dev)
        self.flat32 = torch.zeros(max(n_res32, 1), dtype=F32, device=dev)      # resident masters: root unit + fp32-compute params
        self.flat16 = torch.zeros(max(n_res16, 1), dtype=BF, device=dev)       # resident compute copies (root unit)
        self.g16 = torch.zeros(max(n_res16, 1), dtype=BF, device=dev)
        self.g32 = torch.zeros(max(n_g32, 1), dtype=F32, device=dev)
        self._dummy = torch.zeros(1, dtype=F32, device=dev)
        self.opt_segments: List[Dict] = []
        self.opt_state_size = 0
        self._full_param_numel = sum(p.numel() for p in params)
        o32 = o16 = og32 = 0
        max_n = max([u.n for u, _, _ in plan if u.sharded] + [0])
        for u, lo, hi in plan:
            if u.sharded:
                # the rank's chunk of the unit's range, cut from the (still full) initial parameters
                full = torch.zeros(u.n, dtype=F32, device=dev)
                for p, off, k in u.members:
                    full[off:off + k].copy_(p.data.reshape(-1))
                self.chunk32[u.cs:u.cs + u.ck].copy_(full[self.rank * u.ck:(self.rank + 1) * u.ck])
                del full
                for p, off, k in u.members:
                    p.data = self._dummy.expand(p.shape)       # no full fp32 copy lives on this rank any more
                    p._o2c = None
                    p._o2g = None
                    p._o2_sharded = True
                    p._o2_fresh = True
                    p._o2_engine = self
                    u.params.append(p)
                    self._unit_of[id(p)] = u
                self.opt_segments.append(dict(kind="lo", n=u.ck, os=self.opt_state_size, gather=True, members=u.members,
                                              p32=self.chunk32[u.cs:u.cs + u.ck], g=self.gchunk16[u.cs:u.cs + u.ck],
                                              p16=self.chunk16[u.cs:u.cs + u.ck]))
                self.opt_state_size += u.ck
                # (through __dict__: nn.Module.__setattr__ would register the engine -- a Module holding the whole model -- as a
                # child of the unit)
                u.module.__dict__["_o2_unit_engine"] = self
                u.module.__dict__["_o2_unit"] = u
            elif lo:                                            # resident unit (root): replicated compute copy, sharded update
                s32, s16 = o32, o16
                for p, off, k in u.members:
                    self.flat32[s32 + off:s32 + off + k].copy_(p.data.reshape(-1))
                    p.data = self.flat32[s32 + off:s32 + off + k].view(p.shape)
                    p._o2c = self.flat16[s16 + off:s16 + off + k].view(p.shape)
                    p._o2g = self.g16[s16 + off:s16 + off + k].view(p.shape)
                    p._o2_fresh = True
                    p._o2_engine = self
                    u.params.append(p)
                    self._unit_of[id(p)] = u
                o32 += u.n
                o16 += u.n
                u.res32, u.res16 = s32, s16
                u.grad_views.append(self.g16[s16:s16 + u.n])
                u.lo_view = u.grad_views[-1]
                c0 = self.rank * u.ck
                self.opt_segments.append(dict(kind="lo", n=u.ck, os=self.opt_state_size, gather=True, members=u.members,
                                              p32=self.flat32[s32 + c0:s32 + c0 + u.ck], g=self.g16[s16 + c0:s16 + c0 + u.ck],
                                              p16=self.flat16[s16 + c0:s16 + c0 + u.ck]))
                self.opt_state_size += u.ck
            hi_members, s32h, sg = [], o32, og32
            rep_hi = 0
            for n, p in hi:
                k = p.numel()
                hi_members.append((p, og32 - sg, k))
                if not hasattr(p, "_o2_tp"):
                    rep_hi = og32 - sg + _round_up(k)
                self.flat32[o32:o32 + k].copy_(p.data.reshape(-1))
                p.data = self.flat32[o32:o32 + k].view(p.shape)
                p.grad = self.g32[og32:og32 + k].view(p.shape)
                if p.requires_grad:
                    u.params.append(p)
                    self._unit_of[id(p)] = u
                    p.register_post_accumulate_grad_hook(self._hi_hook)
                o32 += _round_up(k)
                og32 += _round_up(k)
            # gradient ranges of the tensor-parallel replicas on THIS rank (finish_grad_sync broadcasts them)
            u.rep_views = []
            if self.tp_group is not None:
                if u.sharded:
                    a, b = max(0, self.rank * u.ck), min(u.rep_lo, (self.rank + 1) * u.ck)
                    if a < b:
                        u.rep_views.append(self.gchunk16[u.cs + a - self.rank * u.ck:u.cs + b - self.rank * u.ck])
                elif lo and u.rep_lo:
                    u.rep_views.append(self.g16[u.res16:u.res16 + u.rep_lo])
                if rep_hi:
                    u.rep_views.append(self.g32[sg:sg + rep_hi])
            if hi:
                u.grad_views.append(self.g32[sg:og32])
                self.opt_segments.append(dict(kind="hi", n=og32 - sg, os=self.opt_state_size, gather=False, members=hi_members,
                                              p32=self.flat32[s32h:s32h + (og32 - sg)], g=self.g32[sg:og32], p16=None))
                self.opt_state_size += og32 - sg
            self.units.append(u)
        self.sharded_units = [u for u in self.units if u.sharded]
        # pools: gathered bf16 parameters (pool_size buffers) and per-unit gradient staging (2 buffers)
        # The pooled buffers start ZEROED and the ranges of a unit that no kernel writes (alignment padding between members and
        # up to world * 128, members with requires_grad = False) are re-zeroed whenever a buffer passes to a unit of another
        # layout (pre_backward): whatever sits there is reduce-scattered into gchunk16, seen by the scaler's finite check and
        # applied by AdamW, so it must never be allocator garbage (a non-finite pattern would skip every step and decay the loss
        # scale to its floor).
        self.ppool = [torch.zeros(max(max_n, 1), dtype=BF, device=dev) for _ in range(pool_size)]
        self.gpool = [torch.zeros(max(max_n, 1), dtype=BF, device=dev) for _ in range(2)]
        self._glayout = [None, None]                   # layout signature of the unit that last wrote each gradient buffer
        for u in self.sharded_units:
            pos, gaps = 0, []
            for p_, off, k in u.members:
                if off > pos:
                    gaps.append((pos, off))
                if p_.requires_grad:
                    pos = off + k
                else:
                    pos = off                              # a frozen member is a gap: nothing writes its gradient
                    gaps.append((off, off + k))
                    pos = off + k
            if pos < u.n:
                gaps.append((pos, u.n))
            u.gaps = gaps
            u.layout = tuple((off, k, bool(p_.requires_grad)) for p_, off, k in u.members)
        self._pfree = list(range(pool_size))
        self._gfree = [0, 1]
        self._pfree_ev = [None] * pool_size            # compute-stream events: last kernel using the buffer has been queued
        self._gfree_ev = [None, None]                  # communication-stream events: reduce-scatter of the buffer done
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.comm_stats: Optional[CommStats] = None      # set to a CommStats() to account communication (bench, tests)
        self._order: List[_Unit] = []                  # execution order of the sharded units in forward (recorded)
        self._recording = True
        self._launched: List[_Unit] = []
        self.refresh_compute_copies()
        self.zero_grad()

    # ---- sizes ---------------------------------------------------------------------------------------------------------
    def param_bytes_per_rank(self) -> Dict[str, int]:
        """persistent parameter-state bytes on this rank (fp32 master + bf16 compute copy + gradient; the optimizer adds
        8 bytes per element of `opt_state_size`), next to what the replicated engine keeps"""
        sh = sum(u.ck for u in self.sharded_units)
        res_lo = sum(u.n for u in self.units if not u.sharded)
        hi = self.g32.numel()
        return {"sharded_units": 8 * sh, "resident": 8 * res_lo + 8 * hi,
                "transient_pools": 2 * sum(b.numel() for b in self.ppool) + 2 * sum(b.numel() for b in self.gpool),
                "replicated_engine_would_keep": 8 * (sum(u.n for u in self.units) + hi),
                "optimizer_state": 8 * self.opt_state_size}

    # ---- compute copies -----------------------------------------------------------------------------------------------
    def refresh_compute_copies(self):
        def cast(src, dst):
            if src.is_cuda:
                from .. import _hip
                _hip.cast_to_bf16(src, dst)
            else:
                dst.copy_(src)
        if self.sharded_units:
            cast(self.chunk32, self.chunk16)
        for u in self.units:
            if not u.sharded and u.n:
                cast(self.flat32[u.res32:u.res32 + u.n], self.flat16[u.res16:u.res16 + u.n])

    # ---- gather / release of a sharded unit ------------------------------------------------------------------------------
    def _cur(self):
        return torch.cuda.current_stream() if self.comm_stream is not None else None

    def _issue_gather(self, u: _Unit):
        """all-gather the unit's chunks into a pooled buffer (communication stream); no-op if already resident / in flight"""
        if u.pbuf is not None:
            return
        if not self._pfree:
            raise RuntimeError("parameter pool exhausted: a unit was not released (pool_size too small for the prefetch depth)")
        b = self._pfree.pop(0)
        u.pbuf = b
        buf = self.ppool[b][:u.n]
        mine = self.chunk16[u.cs:u.cs + u.ck]

        def run():
            if self.comm:
                dist.all_gather_into_tensor(buf, mine if dist.get_backend(self.pg) == "nccl" else mine.clone(), group=self.pg)
            else:
                buf.copy_(mine)
        if self.comm_stream is not None:
            ev = self._pfree_ev[b]
            if ev is not None:
                self.comm_stream.wait_event(ev)            # the buffer's previous user has finished computing
            self.comm_stream.wait_stream(torch.cuda.current_stream())     # chunk16 is current (AdamW / cast ran on the compute stream)
            cs = self.comm_stats
            with torch.cuda.stream(self.comm_stream):
                if cs is not None:
                    e0, e1 = cs._ev(), cs._ev()
                    e0.record(self.comm_stream)
                run()
                if cs is not None:
                    e1.record(self.comm_stream)
                    cs.spans.append((e0, e1))
                    cs.bytes += buf.numel() * buf.element_size()
                    cs.launches += 1
                u.pevent = torch.cuda.Event()
                u.pevent.record(self.comm_stream)
        else:
            run()

    def _acquire(self, u: _Unit):
        self._issue_gather(u)
        if u.pevent is not None:
            cs = self.comm_stats
            if cs is not None:                       # how long the compute stream waits for the gathered unit
                e0, e1 = cs._ev(), cs._ev()
                e0.record(torch.cuda.current_stream())
                torch.cuda.current_stream().wait_event(u.pevent)
                e1.record(torch.cuda.current_stream())
                cs.stalls.append((e0, e1))
            else:
                torch.cuda.current_stream().wait_event(u.pevent)
            u.pevent = None
        buf = self.ppool[u.pbuf]
        for p, off, k in u.members:
            p._o2c = buf[off:off + k].view(p.shape)

    def _release(self, u: _Unit):
        if u.pbuf is None:
            return
        b = u.pbuf
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._pfree_ev[b] = ev
        for p, _, _ in u.members:
            p._o2c = None
        u.pbuf = None
        self._pfree.append(b)

    def _neighbour(self, u: _Unit, step: int) -> Optional[_Unit]:
        if not self.prefetch or u not in self._order:
            return None
        i = self._order.index(u) + step
        return self._order[i] if 0 <= i < len(self._order) else None

    def pre_forward(self, mod):
        u = mod._o2_unit
        if self._recording and u not in self._order:
            self._order.append(u)
        self._acquire(u)
        nxt = self._neighbour(u, +1)
        if nxt is not None:
            self._issue_gather(nxt)                        # forward_prefetch: the next unit travels while this one computes

    def post_forward(self, mod):
        self._release(mod._o2_unit)

    def pre_backward(self, mod):
        u = mod._o2_unit
        self._recording = False
        self._acquire(u)
        if u.gbuf is None:
            if not self._gfree:
                raise RuntimeError("gradient pool exhausted")
            g = self._gfree.pop(0)
            u.gbuf = g
            if self.comm_stream is not None and self._gfree_ev[g] is not None:
                torch.cuda.current_stream().wait_event(self._gfree_ev[g])      # its last reduce-scatter has finished
            buf = self.gpool[g]
            if self._glayout[g] != u.layout:              # another layout wrote here last: its members overlap this unit's gaps
                for a, b in u.gaps:
                    buf[a:b].zero_()
                self._glayout[g] = u.layout
            for p, off, k in u.members:
                p._o2g = buf[off:off + k].view(p.shape)
                p._o2_fresh = True
        prv = self._neighbour(u, -1)
        if prv is not None:
            self._issue_gather(prv)

    def post_backward(self, mod):
        """the unit's backward has returned (gate on the unit's input): its gathered parameters go back to the pool"""
        self._release(mod._o2_unit)

    # ---- gradient life cycle -----------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False):
        pend = getattr(self.module, "_tables_pending", None)      # (a forward whose backward never ran must not pin the model
        if pend is not None:                                         #  to the ATen table path: a new step starts clean)
            pend[0] = 0
        self.g32.zero_()
        for u in self.units:
            u.pending = sum(1 for p in u.params if p.requires_grad)
            u.handle = None
            u.launched = False
            for p in u.params:
                if hasattr(p, "_o2_fresh"):
                    p._o2_fresh = True
        self._launched = []
        self._replicas_synced = False

    def _hi_hook(self, p):
        self.grad_ready(p)

    def grad_ready(self, p):
        u = self._unit_of.get(id(p))
        if u is None:
            return
        u.pending -= 1
        if u.pending == 0:
            self._launch(u)

    def _reduce_views(self, u: _Unit):
        """(what to run on the communication stream for this unit) -> list of async work handles"""
        hs = []
        nccl = self.comm and dist.get_backend(self.pg) == "nccl"
        if u.sharded:
            buf = self.gpool[u.gbuf][:u.n]
            out = self.gchunk16[u.cs:u.cs + u.ck]
            if self.comm:
                if nccl:
                    hs.append(dist.reduce_scatter_tensor(out, buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                else:                                   # gloo: the all-reduce leaves the same sum in this rank's chunk
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
                    out.copy_(buf[self.rank * u.ck:(self.rank + 1) * u.ck])
                if self.rg is not None:
                    if hs:
                        hs[-1].wait()
                        hs = []
                    hs.append(dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.rg, async_op=True))
            else:
                out.copy_(buf[:u.ck])
        else:
            for v in u.grad_views:
                if self.comm:
                    hs.append(dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                    if self.rg is not None:
                        hs[-1].wait()
                        hs[-1] = dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.rg, async_op=True)
        return hs

    def _launch(self, u: _Unit):
        if u.launched:
            return
        u.launched = True
        self._launched.append(u)
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            cs = self.comm_stats
            with torch.cuda.stream(self.comm_stream):
                if cs is not None:
                    e0, e1 = cs._ev(), cs._ev()
                    e0.record(self.comm_stream)
                u.handle = self._reduce_views(u)
                if u.sharded or cs is not None:
                    for h in u.handle:
                        h.wait()                      # stream-ordered for NCCL work: later communication-stream work follows it
                if cs is not None:
                    e1.record(self.comm_stream)
                    cs.spans.append((e0, e1))
                    cs.bytes += (u.n * 2 if u.sharded else sum(v.numel() * v.element_size() for v in u.grad_views))
                    cs.launches += 1
                if u.sharded:
                    gev = torch.cuda.Event()
                    gev.record(self.comm_stream)
                    self._gfree_ev[u.gbuf] = gev
        else:
            u.handle = self._reduce_views(u)
            for h in u.handle:
                h.wait()
        if u.sharded:                                   # the gradient buffer goes back to the pool (its next user waits for
            g = u.gbuf                                  # the reduce-scatter's event); the gathered parameters stay until
            for p, _, _ in u.members:                   # post_backward: input gradients of the unit may still be due
                p._o2g = None
            u.gbuf = None
            self._gfree.append(g)

    def finish_grad_sync(self):
        for u in self.units:
            if not u.launched and (self.comm or u.sharded):
                if u.sharded and u.gbuf is None:
                    raise RuntimeError("unit %s received no gradient this step: parameter-sharded units must take part in "
                                       "every backward" % u.name)
                self._launch(u)
        for u in self._launched:
            if u.handle:
                for h in u.handle:
                    h.wait()
                u.handle = None
        for u in self.sharded_units:                    # a unit whose input needed no gradient never saw post_backward
            self._release(u)
        if self.comm_stream is not None:
            cs = self.comm_stats
            if cs is not None and self._launched:
                e0, e1 = cs._ev(), cs._ev()
                e0.record(torch.cuda.current_stream())
                torch.cuda.current_stream().wait_stream(self.comm_stream)
                e1.record(torch.cuda.current_stream())
                cs.stalls.append((e0, e1))
            else:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        if self.replica_guard is not None and not self._replicas_synced:
            self.replica_guard.after_reduction(self.replica_grad_views())
            self._replicas_synced = True

    def replica_grad_views(self):
        """reduced gradient ranges (this rank's chunks) of the parameters replicated over the tensor-parallel group"""
        return [v for u in self.units for v in u.rep_views]

    def gather_params(self):
        """after the local AdamW: the resident (root) unit's compute copies are re-assembled from the ranks' chunks; sharded
        units need nothing (their next gather reads the updated chunks)"""
        if not self.comm:
            return
        inplace = dist.get_backend(self.pg) == "nccl"
        for u in self.units:
            if not u.sharded and u.n:
                rng = self.flat16[u.res16:u.res16 + u.n]
                mine = rng[self.rank * u.ck:(self.rank + 1) * u.ck]
                dist.all_gather_into_tensor(rng, mine if inplace else mine.clone(), group=self.pg)

    def gather_range(self, chunk: torch.Tensor) -> torch.Tensor:
        if not self.comm:
            return chunk
        full = torch.empty(chunk.numel() * self.world, dtype=chunk.dtype, device=chunk.device)
        dist.all_gather_into_tensor(full, chunk.contiguous().clone(), group=self.pg)
        return full

    # ---- checkpoints: the full fp32 state dict is assembled / cut unit by unit -----------------------------------------------
    def state_dict(self, *a, offload_to_cpu: bool = False, **k):
        """every rank gets the full fp32 state dict with the reference's key names (collective).  offload_to_cpu: assemble it
        unit by unit onto the HOST (what the driver saves): at most one gathered unit lives on the GPU at a time, instead of 4
        bytes per parameter of the whole model (38 GB for interm_10b) next to the sharded state it was sharded to avoid."""
        sd = self.module.state_dict(*a, **k)
        if offload_to_cpu:
            sd = type(sd)((kk, v.cpu()) for kk, v in sd.items())
        names = {id(p): n for n, p in self.module.named_parameters()}
        for u in self.units:
            if not u.n:
                continue
            if u.sharded:
                full = self.gather_range(self.chunk32[u.cs:u.cs + u.ck])
            else:
                c0 = self.rank * u.ck
                full = self.gather_range(self.flat32[u.res32 + c0:u.res32 + c0 + u.ck].clone())
            for p, off, kk in u.members:
                v = full[off:off + kk].view(p.shape)
                sd[names[id(p)]] = v.cpu() if offload_to_cpu else v.clone()
            del full
        return sd

    def load_state_dict(self, sd, strict: bool = True, **k):
        names = {id(p): n for n, p in self.module.named_parameters()}
        sharded_names = {names[id(p)] for u in self.sharded_units for p, _, _ in u.members}
        rest = {kk: v for kk, v in sd.items() if kk not in sharded_names}
        missing = [n for n in sharded_names if n not in sd]
        if strict and missing:
            raise RuntimeError("missing keys in state_dict: %s" % missing[:5])
        r = self.module.load_state_dict(rest, strict=False)
        if strict:      # nn.Module semantics for the resident keys too (the sharded names are absent from `rest` on purpose)
            miss = [k_ for k_ in r.missing_keys if k_ not in sharded_names]
            if miss or r.unexpected_keys:
                raise RuntimeError("Error(s) in loading state_dict: missing keys %s, unexpected keys %s"
                                   % (miss[:5], list(r.unexpected_keys)[:5]))
        for u in self.sharded_units:
            full = torch.zeros(u.n, dtype=F32, device=self.device)
            cur = self.gather_range(self.chunk32[u.cs:u.cs + u.ck]) if missing else None
            for p, off, kk in u.members:
                t = sd.get(names[id(p)])
                if t is not None:
                    full[off:off + kk].copy_(t.reshape(-1).to(self.device, F32))
                elif cur is not None:
                    full[off:off + kk].copy_(cur[off:off + kk])
            self.chunk32[u.cs:u.cs + u.ck].copy_(full[self.rank * u.ck:(self.rank + 1) * u.ck])
        self.refresh_compute_copies()
        return r

    def consolidate_master(self):
        pass

    # ---- nn.Module surface ---------------------------------------------------------------------------------------------------
    def forward(self, *a, **k):
        return self.module(*a, **k)

    def data_config(self, *a, **k):
        return self.module.data_config(*a, **k)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)
