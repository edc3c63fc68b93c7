"""Head-split tensor parallelism over xGMI for 10b-class models.

The reference splits every transformer Block and the variable-aggregation attention over `tensor_par_size` ranks:
column-parallel `qkv` / `fc1` (heads resp. hidden units are divided), row-parallel `proj` / `fc2`, one all-reduce of
the [B*L, D] partial products after each row-parallel Linear in the forward pass and one all-reduce of the input
gradient before each column-parallel Linear in the backward pass (components/attention.py:36-50,81-85,
components/mlp.py:50-71, utils/dist_functions.py:430-445,533-548).  This module holds the pieces that are not
kernels: the collective, parameter tagging, replicated-parameter sync and the state_dict shard / merge helpers.

Numerics decisions:
  * each rank adds ITS OWN row-parallel bias before the all-reduce, exactly as the reference does
    (`self.proj(x)` then `dist.all_reduce`), so the effective bias is the SUM of the ranks' biases and a reference
    per-rank checkpoint (`<path>_rank_<r>`) loads and evaluates identically;
  * the ranks of one tensor-parallel group draw the same dropout / DropPath seeds (they are seeded by their
    data-parallel rank), so replicated activations stay identical without the reference's repair broadcasts
    (res_slimvit.py:223-226,286-297); attention-probability dropout is decorrelated between ranks by folding
    the tensor-parallel rank into its seed.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

# parameter-name suffix -> (kind, axis) of the reference's split.  "qkv": rows are [3][heads][d] -> the heads axis is cut;
# "kv": rows are [2][heads][d]; "rows": plain row (output-feature) split; "cols": input-feature split.
_SPLIT_RULES = (
    ("attn.qkv.weight", "qkv"), ("attn.qkv.bias", "qkv"), ("attn.proj.weight", "cols"),
    ("mlp.fc1.weight", "rows"), ("mlp.fc1.bias", "rows"), ("mlp.fc2.weight", "cols"),
    ("var_agg.q.weight", "rows"), ("var_agg.kv.weight", "kv"), ("var_agg.proj.weight", "cols"),
)
# row-parallel biases: every rank keeps a full-length vector and the forward SUMS them (see the module docstring)
_SUMMED_BIASES = ("attn.proj.bias", "mlp.fc2.bias", "var_agg.proj.bias")


def split_kind(name: str) -> Optional[str]:
    for suffix, kind in _SPLIT_RULES:
        if name.endswith(suffix):
            return kind
    return None


def group_size(group) -> int:
    return dist.get_world_size(group) if (group is not None and dist.is_initialized()) else 1


def group_rank(group) -> int:
    return dist.get_rank(group) if (group is not None and dist.is_initialized()) else 0


def all_reduce_sum(t: torch.Tensor, group) -> torch.Tensor:
    """In-place SUM over the tensor-parallel group, ordered after / before the caller's work on the current stream
    (RCCL: one ring step per xGMI link for 2 ranks; the [B*L, D] bf16 operand is 50 MB per sample at interm_1b).

    The gloo branch exists so the 2-rank parity test can run two processes on ONE card (RCCL refuses two ranks on the
    same device): it stages the operand through host memory.  It moves bytes only; no arithmetic but the sum."""
    if group is None or group_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.detach().to("cpu", torch.float32)
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host.to(t.dtype))
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_max(t: torch.Tensor, group) -> torch.Tensor:
    """MAX over `group` (None = all ranks); same host staging as all_reduce_sum for the one-card gloo test"""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        t.copy_(host)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def broadcast_first(t: torch.Tensor, group) -> torch.Tensor:
    """in-place broadcast from the group's first rank (host-staged under gloo, see all_reduce_sum)"""
    src = dist.get_global_rank(group, 0)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.detach().cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
        return t
    dist.broadcast(t, src=src, group=group)
    return t


class ReplicaGuard:
    """Keeps the parameters that are REPLICATED over a tensor-parallel group identical on its ranks.

    Their gradients agree bit for bit without any exchange as long as (a) every kernel on their gradient path sums in a fixed
    order and (b) all ranks of the group see the same reduced gradient.  (a) holds for the MFMA variable-aggregation backward,
    not for its scalar fallback (fp32 atomics; head dims other than 64 / 128 / 256, more than 25 variables, or
    ORBIT2_VARAGG_SCALAR -- `_hip.atomics_in_grad_path` records its use).  (b) holds when the group's ranks are the whole job;
    with data parallelism beside it each tensor-parallel column reduces its replicas over a DIFFERENT communicator and GPU set,
    and ring / tree order -- hence the bf16 / fp32 summation order -- need not match between columns.  An ulp of difference in a
    near-zero gradient becomes +-lr under AdamW and the replicas drift apart silently.  Policy (ORBIT2_TP_REPLICA_SYNC):
      auto (default)  broadcast the replica gradient ranges from the group's first rank after the data-parallel reduction
                      whenever (a) or (b) is not guaranteed; otherwise no exchange, but every `check_every`-th step (and the
                      first) a pair of checksums of the ranges (plain and position-weighted) is compared across the group and a mismatch raises;
      broadcast       always broadcast;   check  never broadcast, always compare;   off  neither."""

    def __init__(self, group, data_world: int, check_every: int = 100):
        import os
        self.group, self.data_world, self.check_every = group, int(data_world), int(check_every)
        self.mode = os.environ.get("ORBIT2_TP_REPLICA_SYNC", "auto")
        if self.mode not in ("auto", "broadcast", "check", "off"):
            raise ValueError("ORBIT2_TP_REPLICA_SYNC must be auto, broadcast, check or off")
        self.steps = self.broadcasts = self.checks = 0

    def needs_broadcast(self) -> bool:
        from .. import _hip
        if self.mode == "broadcast":
            return True
        return self.mode == "auto" and (self.data_world > 1 or _hip.atomics_in_grad_path)

    def after_reduction(self, views) -> None:
        """once per step, when the data-parallel reduction of `views` (the replica gradient ranges) has completed on the
        current stream"""
        if self.group is None or self.mode == "off" or not views:
            return
        self.steps += 1
        if self.needs_broadcast():
            for v in views:
                broadcast_first(v, self.group)
            self.broadcasts += 1
            return
        if self.mode == "check" or self.steps == 1 or self.steps % self.check_every == 0:
            # two checksums per range over its raw words w[i], in wrapping int64: sum w[i] and sum (2 i + 1) w[i].  The plain sum
            # alone passes two words that differ by opposite amounts (+1 ulp here, -1 ulp there) and any permutation; the
            # position-weighted one does not (odd weights: a single differing word always changes it; two that cancel in the
            # plain sum change it by delta x 2 (j - i) != 0).  Equal on all ranks <=> max == min.  (advisor, round 5)
            sums = []
            for v in views:
                raw = (v.view(torch.int16) if v.element_size() == 2 else v.view(torch.int32)).reshape(-1)
                s0 = torch.zeros((), dtype=torch.int64, device=raw.device)
                s1 = torch.zeros((), dtype=torch.int64, device=raw.device)
                for a in range(0, raw.numel(), 1 << 24):                       # (bounded temporaries: 16 M words at a time)
                    w = raw[a:a + (1 << 24)].to(torch.int64)
                    s0 += w.sum()
                    s1 += (w * (2 * torch.arange(a, a + w.numel(), dtype=torch.int64, device=raw.device) + 1)).sum()
                sums += [s0, s1]
            mine = torch.stack(sums)
            hi, lo = all_reduce_max(mine.clone(), self.group), -all_reduce_max(-mine.clone(), self.group)
            self.checks += 1
            if not torch.equal(hi, lo):
                bad = [i for i in range(len(views)) if int(hi[2 * i]) != int(lo[2 * i]) or int(hi[2 * i + 1]) != int(lo[2 * i + 1])]
                raise RuntimeError("tensor-parallel replicas disagree: the gradient ranges %s of the parameters replicated over "
                                   "the group differ between its ranks at step %d (a non-reproducible kernel or reduction "
                                   "order on their path); run with ORBIT2_TP_REPLICA_SYNC=broadcast" % (bad, self.steps))


class IdentityFwdAllReduceBwd(torch.autograd.Function):
    """input of a column-parallel Linear: identity forward, SUM of the partial input gradients backward
    (reference `F_Identity_B_AllReduce`, utils/dist_functions.py:430-445)"""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return all_reduce_sum(g.contiguous().clone(), ctx.group), None


class AllReduceFwdIdentityBwd(torch.autograd.Function):
    """output of a row-parallel Linear: SUM of the partial products forward, identity backward
    (reference `F_AllReduce_B_Identity`, utils/dist_functions.py:533-548)"""

    @staticmethod
    def forward(ctx, x, group):
        return all_reduce_sum(x.contiguous().clone(), group)

    @staticmethod
    def backward(ctx, g):
        return g, None


def tag_sharded(module: torch.nn.Module) -> None:
    """marks the parameters that differ between tensor-parallel ranks (`_o2_tp` = split kind)"""
    for name, p in module.named_parameters():
        k = split_kind(name)
        if k is not None:
            p._o2_tp = k


@torch.no_grad()
def sync_replicated(module: torch.nn.Module, group) -> None:
    """Broadcasts every parameter that is NOT split from the group's first rank, the counterpart of the reference's
    `initial_0.pth` hand-off when training from scratch (examples/intermediate_downscaling.py:83-112)."""
    if group_size(group) == 1:
        return
    for name, p in module.named_parameters():
        if split_kind(name) is None and not name.endswith(_SUMMED_BIASES):
            broadcast_first(p.data, group)


# ---------------------------------------------------------------------------------------------------------------------
# state_dict conversion: one full (tensor_par_size = 1) dict  <->  the reference's per-rank dicts
# ---------------------------------------------------------------------------------------------------------------------
def _cut(t: torch.Tensor, kind: str, tp: int, r: int, heads: int) -> torch.Tensor:
    if kind == "cols":
        n = t.shape[1] // tp
        return t[:, r * n:(r + 1) * n].clone()
    if kind == "rows":
        n = t.shape[0] // tp
        return t[r * n:(r + 1) * n].clone()
    parts = 3 if kind == "qkv" else 2
    d = t.shape[0] // (parts * heads)
    v = t.view(parts, heads, d, *t.shape[1:])
    hl = heads // tp
    return v[:, r * hl:(r + 1) * hl].reshape(parts * hl * d, *t.shape[1:]).clone()


def shard_state_dict(full: Dict[str, torch.Tensor], tp: int, rank: int, heads: int) -> Dict[str, torch.Tensor]:
    """The slice of a tensor_par_size=1 state_dict that tensor-parallel rank `rank` of `tp` owns.  Row-parallel
    biases go to rank 0 whole and are zero elsewhere, so that their sum is the original bias."""
    if heads % tp:
        raise ValueError("model heads % tensor parallel size must be 0")
    out = {}
    for k, t in full.items():
        kind = split_kind(k)
        if kind is not None:
            out[k] = _cut(t, kind, tp, rank, heads)
        elif k.endswith(_SUMMED_BIASES):
            out[k] = t.clone() if rank == 0 else torch.zeros_like(t)
        else:
            out[k] = t.clone()
    return out


def merge_state_dicts(shards: Sequence[Dict[str, torch.Tensor]], heads: int) -> Dict[str, torch.Tensor]:
    """Inverse of shard_state_dict for dicts saved by the ranks of one tensor-parallel group (reference layout
    `<path>_rank_<r>`): concatenates the split tensors, SUMS the row-parallel biases, takes rank 0's copy of the rest."""
    tp = len(shards)
    out = {}
    for k, t0 in shards[0].items():
        kind = split_kind(k)
        ts: List[torch.Tensor] = [s[k] for s in shards]
        if kind == "cols":
            out[k] = torch.cat(ts, 1)
        elif kind == "rows":
            out[k] = torch.cat(ts, 0)
        elif kind in ("qkv", "kv"):
            parts = 3 if kind == "qkv" else 2
            hl = heads // tp
            d = t0.shape[0] // (parts * hl)
            out[k] = torch.cat([t.view(parts, hl, d, *t.shape[1:]) for t in ts], 1).reshape(parts * heads * d,
                                                                                            *t0.shape[1:])
        elif k.endswith(_SUMMED_BIASES):
            out[k] = torch.stack(ts).sum(0)
        else:
            out[k] = t0.clone()
    return out
