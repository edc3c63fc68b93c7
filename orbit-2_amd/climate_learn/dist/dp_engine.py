"""Data-parallel engine for the HIP training path: the MI355X counterpart of the reference's
FSDP(NO_SHARD) + MixedPrecision(bf16) wrap (examples/intermediate_downscaling.py:583-621).

One process per GPU.  Parameters are re-homed into flat buffers, grouped into the same units FSDP's
transformer_auto_wrap_policy({Block, Sequential}) would make (one per Block, one per Sequential, one root):

    flat32  fp32 master copy of every parameter (param.data becomes a view -> state_dict stays fp32)
    flat16  bf16 compute copy of the GEMM / LayerNorm parameters       (param._o2c)
    g16     bf16 gradient bucket of those parameters, written by the backward kernels (param._o2g)
    g32     fp32 gradient bucket of the remaining (fp32-compute) parameters (param.grad views)

As soon as every parameter of a unit has its gradient (the backward kernels call grad_ready), the unit's
bucket is all-reduced on a dedicated HIP stream (RCCL over xGMI; `gloo` in CPU tests), ordered after the
compute stream by an event, so communication overlaps the rest of backward.  The sum is divided by the
world size inside the fused AdamW (grad_scale), as FSDP's NO_SHARD gradient averaging does.

`shard_optimizer=True` (the SHARD_GRAD_OP-like step towards the reference's FULL/HYBRID sharding): every
unit's bf16 bucket is REDUCE-SCATTERED instead of all-reduced, each rank runs AdamW on its 1/N chunk only (fp32 master
chunk + both moments: optimizer state and update time shrink by N) and the updated bf16 compute copies are ALL-GATHERED
back; same bytes on the wire as the all-reduce.  The fp32 masters of the other ranks' chunks are refreshed on demand
(`consolidate_master`, called by `state_dict`).  The small fp32-compute parameters stay replicated.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import tp as _tp

BF, F32 = torch.bfloat16, torch.float32
_ALIGN = 128  # elements; keeps every view 256-byte aligned


def _round_up(n, a=_ALIGN):
    return (n + a - 1) // a * a


def default_units(module: nn.Module, unit_types: Tuple[type, ...]) -> List[Tuple[str, List[Tuple[str, nn.Parameter]]]]:
    """[(unit name, [(param name, param)])]: one unit per instance of unit_types, the rest in 'root'."""
    claimed = set()
    units = []
    for name, sub in module.named_modules():
        if name and isinstance(sub, unit_types):
            ps = [(name + "." + n, p) for n, p in sub.named_parameters() if id(p) not in claimed]
            if ps:
                for _, p in ps:
                    claimed.add(id(p))
                units.append((name, ps))
    root = [(n, p) for n, p in module.named_parameters() if id(p) not in claimed]
    if root:
        units.append(("root", root))
    return units


class Bucket:
    def __init__(self, name):
        self.name = name
        self.params: List[nn.Parameter] = []
        self.grad_views: List[torch.Tensor] = []   # contiguous ranges to all-reduce (bf16 and/or fp32)
        self.lo_view: Optional[torch.Tensor] = None   # the bf16 range (reduce-scattered when the optimizer is sharded)
        self.pending = 0
        self.handle = None
        self.event = None
        self.rep_views: List[torch.Tensor] = []    # gradient ranges of the parameters replicated over a tensor-parallel group


class CommStats:
    """Per-step communication accounting of an engine (bench.py --gpus N, tests): HIP-event spans on the communication
    stream around every bucket collective (`comm_ms`: how long the stream was busy with them), the compute stream's stall at
    the join in finish_grad_sync / at a gathered unit's hand-over (`exposed_ms`: communication that did NOT hide behind
    compute) and the bytes handed to the collectives.  Off unless `engine.comm_stats` is set: the collectives are then
    waited for on the communication stream right after their launch (stream-ordered for RCCL: nothing blocks the host or the
    compute stream), so that an end event can be recorded behind them."""

    def __init__(self):
        self.spans, self.stalls = [], []
        self.bytes = 0
        self.launches = 0

    @staticmethod
    def _ev():
        return torch.cuda.Event(enable_timing=True)

    def summary(self, steps: int = 1):
        torch.cuda.synchronize()
        comm = sum(a.elapsed_time(b) for a, b in self.spans)
        exposed = sum(a.elapsed_time(b) for a, b in self.stalls)
        return {"comm_ms_per_step": comm / steps, "exposed_comm_ms_per_step": exposed / steps,
                "comm_bytes_per_step": self.bytes / steps, "collectives_per_step": self.launches / steps}


class HipDataParallel(nn.Module):

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
    def __init__(self, module: nn.Module, process_group=None, unit_types: Tuple[type, ...] = (),
                 is_lowp=None, sync_module_states: bool = True, overlap: bool = True,
                 shard_optimizer: bool = False, replica_group=None):
        """replica_group: the tensor-parallel group of this rank (dist/tp.py).  The parameters that are NOT split over it are
        replicas; their gradient ranges are laid out first inside every unit (`replica_grad_views()`).  Where every kernel on
        their gradient path sums in a fixed order AND the group is the whole job, the replicas' gradients agree bit for bit and
        nothing is exchanged (a periodic exact checksum across the group verifies it); with data parallelism beside the group
        (each column reduces over its own communicator) or an atomics-accumulating kernel on the path, the ranges are broadcast
        from the group's first rank after the reduction: dist/tp.py ReplicaGuard."""
        super().__init__()
        self.module = module
        self.pg = process_group
        self.replica_group = replica_group if (replica_group is not None and dist.is_initialized()
                                               and dist.get_world_size(replica_group) > 1) else None
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.replica_guard = _tp.ReplicaGuard(self.replica_group, self.world) if self.replica_group is not None else None
        self._replicas_synced = False
        self.shard = bool(shard_optimizer)
        # ORBIT2_FORCE_COLLECTIVES=1 issues the bucket all-reduces even on a single rank (exercises the RCCL path
        # -- streams, async handles, bf16 reduction -- on a 1-GPU box)
        import os as _os
        self.force_comm = dist.is_initialized() and _os.environ.get("ORBIT2_FORCE_COLLECTIVES", "0") == "1"
        self.overlap = overlap
        params = [p for p in module.parameters()]
        assert params, "module has no parameters"
        dev = params[0].device
        self.device = dev
        if is_lowp is None:
            is_lowp = lambda name, p: getattr(p, "_o2_lowp", False)
        units = default_units(module, unit_types)
        # ---- layout
        n32 = n16 = ng32 = 0
        plan = []
        for uname, ps in units:
            lo = [(n, p) for n, p in ps if is_lowp(n, p)]
            hi = [(n, p) for n, p in ps if not is_lowp(n, p)]
            if self.replica_group is not None:      # replicas first, tensor-parallel shards after (stable order)
                lo.sort(key=lambda np_: hasattr(np_[1], "_o2_tp"))
                hi.sort(key=lambda np_: hasattr(np_[1], "_o2_tp"))
            plan.append((uname, lo, hi))
            ulo = sum(_round_up(p.numel()) for _, p in lo)
            if self.shard:
                ulo = _round_up(ulo, self.world * _ALIGN)       # every rank's chunk of the unit stays aligned
            n16 += ulo
            n32 += ulo + sum(_round_up(p.numel()) for _, p in hi)
            for _, p in hi:
                ng32 += _round_up(p.numel())
        self.flat32 = torch.zeros(n32, dtype=F32, device=dev)
        self.flat16 = torch.zeros(max(n16, 1), dtype=BF, device=dev)
        self.g16 = torch.zeros(max(n16, 1), dtype=BF, device=dev)
        self.g32 = torch.zeros(max(ng32, 1), dtype=F32, device=dev)
        # master layout: per unit [lowp params..., fp32-compute params...]; lowp masters of ALL units are also
        # laid out so that flat16/g16 offsets follow the same order -> AdamW runs on (few) long ranges.
        self.buckets: List[Bucket] = []
        self.lowp_ranges: List[Tuple[int, int, int]] = []   # (off32, off16, n) per unit
        self.hi_ranges: List[Tuple[int, int, int]] = []     # (off32, offg32, n) per unit
        # what the optimizer touches on THIS rank: dicts(kind, o32, og, n, os); os = offset into the moment buffers
        self.opt_segments: List[Dict] = []
        self.opt_state_size = 0
        o32 = o16 = og32 = 0
        self._bucket_of: Dict[int, Bucket] = {}
        for uname, lo, hi in plan:
            bk = Bucket(uname)
            s32, s16 = o32, o16
            rep16 = rep32 = 0                       # length of the replicated prefix of the bf16 / fp32 range
            lo_members, hi_members = [], []         # (param, offset inside the unit's range, numel): checkpoint layout
            for n, p in lo:
                k = p.numel()
                lo_members.append((p, o16 - s16, k))
                if not hasattr(p, "_o2_tp"):
                    rep16 = o16 - s16 + _round_up(k)
                self.flat32[o32:o32 + k].copy_(p.data.reshape(-1))
                p.data = self.flat32[o32:o32 + k].view(p.shape)
                p._o2c = self.flat16[o16:o16 + k].view(p.shape)
                p._o2g = self.g16[o16:o16 + k].view(p.shape)
                p._o2_fresh = True
                p._o2_engine = self
                bk.params.append(p)
                self._bucket_of[id(p)] = bk
                o32 += _round_up(k)
                o16 += _round_up(k)
            if lo:
                if self.shard:
                    pad = _round_up(o16 - s16, self.world * _ALIGN) - (o16 - s16)
                    o16 += pad
                    o32 += pad
                n_lo = o16 - s16
                self.lowp_ranges.append((s32, s16, n_lo))
                bk.grad_views.append(self.g16[s16:o16])
                bk.lo_view = bk.grad_views[-1]
                if self.shard:
                    ck = n_lo // self.world
                    self.opt_segments.append(dict(kind="lo", o32=s32 + self.rank * ck, og=s16 + self.rank * ck, n=ck,
                                                  os=self.opt_state_size, gather=True, members=lo_members))
                    self.opt_state_size += ck
                else:
                    self.opt_segments.append(dict(kind="lo", o32=s32, og=s16, n=n_lo, os=s32, gather=False,
                                                  members=lo_members))
            s32h, sg = o32, og32
            for n, p in hi:
                k = p.numel()
                hi_members.append((p, og32 - sg, k))
                if not hasattr(p, "_o2_tp"):
                    rep32 = og32 - sg + _round_up(k)
                self.flat32[o32:o32 + k].copy_(p.data.reshape(-1))
                p.data = self.flat32[o32:o32 + k].view(p.shape)
                p.grad = self.g32[og32:og32 + k].view(p.shape)
                p._o2_engine = self                # (fused table kernels write such gradients themselves and call grad_ready)
                if p.requires_grad:
                    bk.params.append(p)
                    self._bucket_of[id(p)] = bk
                    p.register_post_accumulate_grad_hook(self._hi_hook)
                o32 += _round_up(k)
                og32 += _round_up(k)
            if hi:
                self.hi_ranges.append((s32h, sg, og32 - sg))
                bk.grad_views.append(self.g32[sg:og32])
                if self.shard:
                    self.opt_segments.append(dict(kind="hi", o32=s32h, og=sg, n=og32 - sg, os=self.opt_state_size,
                                                  gather=False, members=hi_members))
                    self.opt_state_size += og32 - sg
                else:
                    self.opt_segments.append(dict(kind="hi", o32=s32h, og=sg, n=og32 - sg, os=s32h, gather=False,
                                                  members=hi_members))
            if self.replica_group is not None:
                bk.rep_views = [v for v in (self.g16[s16:s16 + rep16], self.g32[sg:sg + rep32]) if v.numel()]
            self.buckets.append(bk)
        if not self.shard:
            self.opt_state_size = n32                      # moments laid out like flat32
        # what the optimizer / loss scaler touch, as tensors (the same dict shape as the parameter-sharding engine's):
        # p32 fp32 master range, g gradient range, p16 bf16 compute range (None for fp32-compute parameters)
        for sg_ in self.opt_segments:
            lo_ = sg_["kind"] == "lo"
            sg_["p32"] = self.flat32[sg_["o32"]:sg_["o32"] + sg_["n"]]
            sg_["g"] = (self.g16 if lo_ else self.g32)[sg_["og"]:sg_["og"] + sg_["n"]]
            sg_["p16"] = self.flat16[sg_["og"]:sg_["og"] + sg_["n"]] if lo_ else None
        self.grad_world = self.world                       # number of ranks whose gradients are summed (AdamW divides)
        self.refresh_compute_copies()
        if sync_module_states and (self.world > 1 or self.force_comm):
            dist.broadcast(self.flat32, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0,
                           group=self.pg)
            self.refresh_compute_copies()
        self.comm_stream = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and overlap) else None
        self.comm_stats: Optional[CommStats] = None      # set to a CommStats() to account communication (bench, tests)
        self._launched: List[Bucket] = []
        self._arrived = set()
        self.zero_grad()

    # ---- parameters <-> compute copies --------------------------------------------------------------
    def refresh_compute_copies(self):
        """flat16 <- bf16(flat32) for every low-precision range (after init / checkpoint load)."""
        for o32, o16, n in self.lowp_ranges:
            if self.flat32.is_cuda:
                from .. import _hip
                _hip.cast_to_bf16(self.flat32[o32:o32 + n], self.flat16[o16:o16 + n])
            else:
                self.flat16[o16:o16 + n].copy_(self.flat32[o32:o32 + n])

    # ---- gradient life cycle ------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False):
        """Logical zero: bf16 buckets are overwritten (beta = 0) by the first backward kernel that touches
        them; the fp32 bucket is accumulated into by autograd, so it is memset."""
        pend = getattr(self.module, "_tables_pending", None)      # (a forward whose backward never ran must not pin the model
        if pend is not None:                                         #  to the ATen table path: a new step starts clean)
            pend[0] = 0
        self.g32.zero_()
        for bk in self.buckets:
            bk.pending = sum(1 for p in bk.params if p.requires_grad)
            bk.handle = None
            for p in bk.params:
                if hasattr(p, "_o2g"):
                    p._o2_fresh = True
        self._launched = []
        self._arrived = set()
        self._replicas_synced = False

    def _hi_hook(self, p):
        self.grad_ready(p)

    def grad_ready(self, p):
        """a parameter's gradient of this step is complete in its bucket.  Idempotent per step: a kernel that writes a gradient
        itself announces it here, and autograd's post-accumulate hook may announce the same parameter again (it fires even when
        the Function returned no gradient for it) -- counted twice, the bucket would be reduced before its last gradients exist"""
        bk = self._bucket_of.get(id(p))
        if bk is None or id(p) in self._arrived:
            return
        self._arrived.add(id(p))
        bk.pending -= 1
        if bk.pending == 0:
            self._launch(bk)

    def _reduce(self, bk: Bucket, v: torch.Tensor):
        """all-reduce, or -- for the bf16 range of a sharded-optimizer engine -- reduce-scatter into this rank's
        chunk of the same buffer (in place: output = input + rank*chunk, the form RCCL runs without a copy)"""
        if self.shard and v is bk.lo_view and dist.get_backend(self.pg) == "nccl":
            ck = v.numel() // self.world
            return dist.reduce_scatter_tensor(v[self.rank * ck:(self.rank + 1) * ck], v, op=dist.ReduceOp.SUM,
                                              group=self.pg, async_op=True)
        # gloo (CPU tests) has no in-place reduce-scatter: the all-reduce leaves the same sum in this rank's chunk
        return dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _launch(self, bk: Bucket):
        self._launched.append(bk)
        if self.world == 1 and not self.force_comm:
            return
        cs = self.comm_stats
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                if cs is not None:
                    e0, e1 = cs._ev(), cs._ev()
                    e0.record(self.comm_stream)
                bk.handle = [self._reduce(bk, v) for v in bk.grad_views]
                if cs is not None:
                    for h in bk.handle:
                        h.wait()                      # stream-ordered: the communication stream (only) waits for the collective
                    e1.record(self.comm_stream)
                    cs.spans.append((e0, e1))
        else:
            bk.handle = [self._reduce(bk, v) for v in bk.grad_views]
        if cs is not None:
            cs.bytes += sum(v.numel() * v.element_size() for v in bk.grad_views)
            cs.launches += len(bk.grad_views)

    def _gather_ranges(self, flat: torch.Tensor, which: int):
        """all-gather every unit's low-precision range of `flat` (flat16: which = 1, flat32: which = 0) from the
        ranks' chunks, in place"""
        if self.world == 1 and not self.force_comm:
            return
        inplace = dist.get_backend(self.pg) == "nccl"
        for rg in self.lowp_ranges:
            off, n = rg[which], rg[2]
            ck = n // self.world
            mine = flat[off + self.rank * ck: off + (self.rank + 1) * ck]
            dist.all_gather_into_tensor(flat[off:off + n], mine if inplace else mine.clone(), group=self.pg)

    def gather_range(self, chunk: torch.Tensor) -> torch.Tensor:
        """the full range a rank-chunk belongs to (ranks' chunks concatenated in rank order); used for checkpoints"""
        if self.world == 1 and not self.force_comm:
            return chunk
        full = torch.empty(chunk.numel() * self.world, dtype=chunk.dtype, device=chunk.device)
        dist.all_gather_into_tensor(full, chunk.contiguous().clone(), group=self.pg)
        return full

    def gather_params(self):
        """sharded optimizer: after the local AdamW, collect every rank's updated bf16 compute chunk"""
        if self.shard:
            self._gather_ranges(self.flat16, 1)

    def consolidate_master(self):
        """sharded optimizer: refresh the fp32 masters of the chunks other ranks own (checkpointing)"""
        if self.shard:
            self._gather_ranges(self.flat32, 0)

    def finish_grad_sync(self):
        """Block the compute stream until every launched all-reduce is done; reduce stragglers (units whose
        parameters did not all receive a gradient this step)."""
        for bk in self.buckets:
            if bk not in self._launched and bk.pending != sum(1 for p in bk.params if p.requires_grad):
                self._launch(bk)      # partially-ready unit: reduce what is there
            elif bk not in self._launched and (self.world > 1 or self.force_comm):
                self._launch(bk)      # untouched unit still has to take part in the collective
        for bk in self._launched:
            if bk.handle:
                for h in bk.handle:
                    h.wait()
                bk.handle = None           # idempotent: a second call (e.g. scaler.step after a captured step) is a no-op
        if self.comm_stream is not None:
            cs = self.comm_stats
            if cs is not None and self._launched and (self.world > 1 or self.force_comm):
                e0, e1 = cs._ev(), cs._ev()
                e0.record(torch.cuda.current_stream())          # the last backward kernel has been queued
                torch.cuda.current_stream().wait_stream(self.comm_stream)
                e1.record(torch.cuda.current_stream())          # ... e1 - e0 = what the compute stream waited for communication
                cs.stalls.append((e0, e1))
            else:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        if self.replica_guard is not None and not self._replicas_synced:
            self.replica_guard.after_reduction(self.replica_grad_views())
            self._replicas_synced = True

    def replica_grad_views(self):
        """gradient ranges of the parameters replicated over the tensor-parallel group (tests: equal on every rank of it)"""
        return [v for bk in self.buckets for v in bk.rep_views]

    # ---- nn.Module surface -----------------------------------------------------------------------------
    def forward(self, *a, **k):
        return self.module(*a, **k)

    def data_config(self, *a, **k):
        return self.module.data_config(*a, **k)

    def state_dict(self, *a, **k):
        self.consolidate_master()
        return self.module.state_dict(*a, **k)

    def load_state_dict(self, sd, *a, **k):
        r = self.module.load_state_dict(sd, *a, **k)
        self.refresh_compute_copies()
        return r

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)
