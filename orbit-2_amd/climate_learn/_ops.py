"""Autograd functions that orchestrate the HIP kernels (climate_learn._hip) for the Res_Slim_ViT hot path.

Each Function is one fused stage of the training step with a hand-written backward made of the same C-ABI
kernels: every GEMM of the step, forward and backward, runs in liborbit2_hip.so (`orbit2_gemm_bf16` /
`orbit2_gemm_bf16_grouped`); no vendor-library GEMM is called anywhere in the package.  Precision policy = the reference's FSDP MixedPrecision
(examples/intermediate_downscaling.py:601-607): fp32 master parameters, bf16 compute copies and activations,
fp32 accumulation inside the kernels, bf16 weight gradients.

Parameters are passed to the Functions as the module's fp32 nn.Parameters (for graph connectivity); the kernels
read `param._o2c` (bf16 compute copy, maintained by the DP engine / optimizer) and write weight gradients
straight into `param._o2g` (a view of the engine's flat gradient bucket), then notify the engine so the
bucket's RCCL all-reduce can start while backward continues.  Without an engine (unit tests) the compute
copy is cast on the fly and gradients are returned to autograd as ordinary tensors.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from . import _hip
from .dist import tp as _tp

_Q_PRESCALE = 1.4426950408889634      # log2(e): q is stored as q * log2(e)/sqrt(d) (include/orbit2_hip.h, ORBIT2_ATTN_Q_PRESCALED)

BF, F32 = torch.bfloat16, torch.float32


# ------------------------------------------------------------------------------------------------------
# dropout seed stream (counter-based: every dropout site of every step gets its own 64-bit seed)
# ------------------------------------------------------------------------------------------------------
class _SeedStream:
    def __init__(self):
        self.base = 0x243F6A8885A308D3
        self.counter = 0

    def manual_seed(self, seed: int, rank: int = 0):
        self.base = (int(seed) * 0x9E3779B97F4A7C15 + (rank + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        self.counter = 0

    def mark(self) -> int:
        """position of the stream (GraphedTrainStep rewinds to it so warm-up and capture issue the same seeds)"""
        return self.counter

    def reset(self, mark: int):
        self.counter = mark

    def next(self) -> int:
        self.counter += 1
        z = (self.base + self.counter * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)


seeds = _SeedStream()


def tp_seed(seed: int, tp_rank: int) -> int:
    """The ranks of a tensor-parallel group share one seed stream (replicated activations get identical masks);
    attention-probability dropout acts on DIFFERENT heads on each rank, so its seed is decorrelated here."""
    return (seed ^ (tp_rank * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def manual_seed(seed: int, rank: int = 0):
    seeds.manual_seed(seed, rank)


# ------------------------------------------------------------------------------------------------------
# parameter plumbing
# ------------------------------------------------------------------------------------------------------
def cw(p: torch.Tensor) -> torch.Tensor:
    """bf16 compute copy of a parameter."""
    c = getattr(p, "_o2c", None)
    if c is not None:
        return c
    if getattr(p, "_o2_sharded", False):
        raise RuntimeError("parameter of a sharded unit used while its unit is not gathered (dist/fsdp_engine.py)")
    return _hip.cast_to_bf16(p.detach().contiguous())


# ------------------------------------------------------------------------------------------------------
# unit scopes: a parameter-sharding engine (dist/fsdp_engine.py, the reference's FSDP FULL / HYBRID_SHARD) gathers a
# unit's bf16 parameters right before the unit's forward and again right before its backward.  Modules whose parameters
# form a unit bracket their fused op with unit_enter / unit_exit; without such an engine both are free.
# ------------------------------------------------------------------------------------------------------
class _UnitBackwardGate(torch.autograd.Function):
    """identity.  On a unit's OUTPUT (last = False) its backward runs before the unit's own backward (it sits downstream of
    the unit in the graph): the engine gathers the unit.  On the unit's INPUT (last = True) it runs after the unit's
    backward has returned: the engine takes the gathered parameters back."""

    @staticmethod
    def forward(ctx, x, mod, last):
        ctx.mod, ctx.last = mod, last
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        eng = getattr(ctx.mod, "_o2_unit_engine", None)
        if eng is not None:
            (eng.post_backward if ctx.last else eng.pre_backward)(ctx.mod)
        return g, None, None


def unit_enter(mod, x):
    """call with the unit's input right before its fused op; returns the input to use"""
    eng = getattr(mod, "_o2_unit_engine", None)
    if eng is None:
        return x
    eng.pre_forward(mod)
    return _UnitBackwardGate.apply(x, mod, True) if torch.is_grad_enabled() and x.requires_grad else x


def unit_exit(mod, out):
    eng = getattr(mod, "_o2_unit_engine", None)
    if eng is None:
        return out
    eng.post_forward(mod)
    return _UnitBackwardGate.apply(out, mod, False) if torch.is_grad_enabled() and out.requires_grad else out


class _GradSink:
    """Where a weight gradient goes: the engine's bucket view (managed) or a fresh tensor handed to autograd."""
    __slots__ = ("param", "buf", "beta", "managed")

    def __init__(self, param: torch.Tensor, dtype=BF):
        g = getattr(param, "_o2g", None)
        self.param = param
        if g is not None:
            self.buf, self.managed = g, True
            self.beta = 0.0 if getattr(param, "_o2_fresh", True) else 1.0
        else:
            self.buf = torch.empty(param.shape, dtype=dtype, device=param.device)
            self.beta, self.managed = 0.0, False

    def done(self):
        """returns what the Function's backward must return for this parameter"""
        if self.managed:
            self.param._o2_fresh = False
            eng = getattr(self.param, "_o2_engine", None)
            if eng is not None:
                eng.grad_ready(self.param)
            return None
        return self.buf


def _needs(ctx, i):
    return ctx.needs_input_grad[i]


# GELU layers: True = the forward epilogue stores GELU'(pre) x dropout factor (save_dact) and the backward multiplies by it (mul);
# False = round 2's form (the forward stores the pre-activation, the backward epilogue evaluates GELU' and the mask again)
_DACT = True


def _dact_ok(p):
    """the q14 factor tensor holds [-2, 2): GELU' (<= 1.13) x the dropout scale 256 / (256 - round(256 p)) must stay inside
    (orbit2_gemm_bf16 rejects save_dact otherwise); above drop_p ~ 0.43 the pre-activation form is used instead"""
    return _DACT and 1.13 * 256.0 / (256.0 - int(p * 256.0 + 0.5)) < 2.0


def _gelu_fwd_kw(buf, p, seed):
    return dict(act=1, save_dact=buf, drop_p=p, seed=seed) if _dact_ok(p) else dict(act=1, save_pre=buf, drop_p=p, seed=seed)


def _gelu_bwd_kw(buf, p, seed):
    return dict(mul=buf) if _dact_ok(p) else dict(drop_p=p, seed=seed, dgelu_pre=buf)


def _ld_pad(n: int) -> int:
    """row pitch (elements) of a [rows, n] bf16 activation that only GEMMs and column sums touch: rows a multiple of 8 KiB apart
    put the same k-offset of EVERY row on the same memory channel (the MLP hidden tensors of interm_1b: 24 KiB rows) and the
    LDS-DMA pieces of a GEMM, 4-8 rows each, queue on it: -13 % on the weight-gradient GEMMs that read them K-strided, -20 % on
    a K-contiguous 4-wave GEMM (profiles/r03_gemm_ld_pad.txt).  128 bytes of padding per row spread them.
    Round 6: the D-wide operands too (LayerNorm outputs, attention output: rows a multiple of 2 KiB apart, e.g. 6 KiB at D = 3072).
    They are the N-side operand of the weight-gradient launch, whose 256-column strips are then served at different rates by the
    memory side (strips at even multiples of 512 B sweep 3-5 % slower than their odd neighbours: profiles/r06_w4_trace_pace0.txt)
    and a cohort's workgroups drift apart; padded, the launch is 1.8 % faster in the step and nothing else moves
    (profiles/r06_ld_pad_small_instep.txt; padding the 3 D-wide qkv rows as well gave it back elsewhere: r06_ld_pad_instep.txt)."""
    if _LD_PAD_SMALL and n < 4096 and (2 * n) % 2048 == 0:
        return n + 64
    return n + 64 if (2 * n) % _LD_PAD_MOD == 0 else n


import os as _os
_LD_PAD_MOD = int(_os.environ.get("ORBIT2_LD_PAD_MOD", "8192"))
_LD_PAD_SMALL = _os.environ.get("ORBIT2_LD_PAD_SMALL", "1") == "1"


def _ld(t, n):
    """leading dimension of a row-major [rows, n] operand (a padded _rows view, or a contiguous tensor of any rank)"""
    return t.stride(0) if t.dim() == 2 else n


def _rows(M, N, device, dtype=BF):
    """[M, N] bf16 (or int16: a GELU-backward factor tensor) with the row pitch of _ld_pad (a view when padded: every consumer
    takes the leading dimension from stride(0))"""
    ld = _ld_pad(N)
    buf = torch.empty(M, ld, dtype=dtype, device=device)
    return buf if ld == N else buf[:, :N]


def _gelu_saved_dtype(p):
    """what a GELU layer's forward keeps for its backward: the q14 fixed-point factor GELU'(pre) x dropout factor (int16,
    save_dact / mul) or, beyond that format's range, the bf16 pre-activation (save_pre / dgelu_pre)"""
    return torch.int16 if _dact_ok(p) else BF


def _linear_fwd(x2d, W, b, M, N, K, pad=False, **kw):
    out = _rows(M, N, x2d.device) if pad else torch.empty(M, N, dtype=BF, device=x2d.device)
    return _hip.gemm(x2d, cw(W), out, M, N, K, _ld(x2d, K), K, out.stride(0), bias=None if b is None else cw(b), **kw)


def _dx(dy2d, W, M, N, K, pad=False, **kw):
    """dx[M,K] = dy[M,N] . W[N,K] on the weight AS STORED: W is the K-strided operand, read through the
    hardware-transposing LDS path of the 8-phase kernel (as fast as the K-contiguous form on a transposed copy --
    profiles/r02_gemm_t8_ab.txt -- so the per-step transposed weight copies of round 1 are gone)."""
    out = _rows(M, K, dy2d.device) if pad else torch.empty(M, K, dtype=BF, device=dy2d.device)
    return _hip.gemm(dy2d, cw(W), out, M, K, N, _ld(dy2d, N), K, out.stride(0), a_kc=True, b_kc=False, **kw)


_FUSE_COLSUM = True   # fc1's bias gradient from the factor-multiply GEMM's per-tile-row sums (bench.py --no-fused-colsum: A/B)
_DW_SPLIT = 8     # K-split of a weight gradient whose output is too few tiles to fill the chip (0: off)


def _dw_split_ok(M, N, K):
    """a lone weight gradient with < 192 tiles of 256 x 256 (the head's 192 x D output layer: 12; a D x D layer at D = 3072: 144)
    and a long contraction (tokens): alone it runs on 48-144 workgroups for 2048 K-tiles; split over the tokens into _DW_SPLIT
    partial products (ONE grouped launch, bf16 partials summed in fp32 in a fixed order: deterministic) it fills the chip"""
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    return _DW_SPLIT > 1 and tiles < 192 and M >= 32768 and M % (_DW_SPLIT * 64) == 0 and N % 8 == 0 and K % 8 == 0


def _dw(dy2d, x2d, W, b, M, N, K):
    """dW[N,K] = dy^T . x ; db[N] = colsum(dy).  Returns what backward must return for (W, b)."""
    sw = _GradSink(W)
    # (an fp32 sink keeps the un-split path: the split's bf16 partials would round an fp32 gradient to 2^-9)
    if _dw_split_ok(M, N, K) and dy2d.dim() == 2 and x2d.dim() == 2 and sw.buf.dtype == BF:
        S, Mc = _DW_SPLIT, M // _DW_SPLIT
        parts = torch.empty(S, N, K, dtype=BF, device=dy2d.device)     # bf16 partials, summed in fp32 by orbit2_batch_sum
        _hip.gemm_grouped([(dy2d[i * Mc:(i + 1) * Mc], x2d[i * Mc:(i + 1) * Mc], parts[i], N, K, Mc, dy2d.stride(0), x2d.stride(0), K,
                            dict(a_kc=False, b_kc=False)) for i in range(S)])
        _hip.batch_sum(parts, S, N, K, sw.buf, beta=sw.beta)
    else:
        _hip.gemm(dy2d, x2d, sw.buf, N, K, M, _ld(dy2d, N), _ld(x2d, K), K, a_kc=False, b_kc=False, beta=sw.beta)
    gw = sw.done()
    gb = None
    if b is not None:
        sb = _GradSink(b)
        _hip.colsum(dy2d, M, N, _ld(dy2d, N), sb.buf, beta=sb.beta)
        gb = sb.done()
    return gw, gb


_DW_BALANCE = int(_os.environ.get("ORBIT2_DW_BALANCE", "4"))      # K-split of the tail tiles of a grouped weight-gradient launch (0: off)


def _dw_balance_plan(shapes, S, slots=256):
    """shapes: [(tiles_m, tiles_n)] of the group's problems.  A grouped launch of T tiles on `slots` workgroup slots (one 256 x 256
    tile per CU) takes ceil(T / slots) sweeps of the contraction; the last one runs T % slots tiles on a partly idle chip (the
    Block's four weight gradients at interm_1b: 1728 tiles = 6.75 rounds, 7 sweeps paid).  Plan: R = T % slots tiles leave the
    full-length set -- whole problems, smallest first, then whole tile rows of the next one -- and are split S ways over the
    tokens: F = T - R full tiles fill whole rounds, the R x S part-length units (R x S a multiple of `slots`) fill whole rounds of
    1 / S the length.  Returns [(problem index, first tile row, tile rows)] to split, or None when no exact plan exists."""
    T = sum(tm * tn for tm, tn in shapes)
    R = T % slots
    if S < 2 or R == 0 or T < slots or (R * S) % slots:
        return None
    order = sorted(range(len(shapes)), key=lambda i: shapes[i][0] * shapes[i][1])
    plan, need = [], R
    for i in order:
        tm, tn = shapes[i]
        if need == 0:
            break
        if tm * tn <= need:
            plan.append((i, 0, tm))
            need -= tm * tn
        elif need % tn == 0:
            r = need // tn
            plan.append((i, tm - r, r))          # the LAST tile rows of the problem
            need = 0
    return plan if need == 0 else None


def _dw_balance(problems):
    """(problems for ONE grouped launch, [(parts, S, rows, K, destination rows, beta)] to sum afterwards): see _dw_balance_plan.
    Applies to whole-tile bf16 problems over one long token range; anything else is returned unchanged."""
    S = _DW_BALANCE
    if S < 2 or len(problems) < 2:
        return problems, []
    M = problems[0][5]
    ok = all(p[5] == M and p[3] % 256 == 0 and p[4] % 256 == 0 and p[2].dtype == BF and p[0].dim() == 2 and p[1].dim() == 2
             for p in problems) and M % (S * 64) == 0 and M // S >= 32768
    if not ok:
        return problems, []
    plan = _dw_balance_plan([(p[3] // 256, p[4] // 256) for p in problems], S)
    if plan is None:
        return problems, []
    full, split, sums = [], [[] for _ in range(S)], []
    cut = {i: (r0, r) for i, r0, r in plan}
    Mc = M // S
    for i, (dy, x, out, N, K, M_, lda, ldb, ldc, kw) in enumerate(problems):
        if i not in cut:
            full.append(problems[i])
            continue
        r0, r = cut[i]
        n0, n1 = 256 * r0, 256 * (r0 + r)
        if r0 > 0:                                   # the problem's first tile rows stay full-length
            full.append((dy[:, :n0], x, out[:n0], n0, K, M_, lda, ldb, ldc, kw))
        parts = torch.empty(S, n1 - n0, K, dtype=BF, device=dy.device)      # bf16 partials, summed in fp32 by orbit2_batch_sum
        for q in range(S):
            split[q].append((dy[q * Mc:(q + 1) * Mc, n0:n1], x[q * Mc:(q + 1) * Mc], parts[q], n1 - n0, K, Mc, lda, ldb, K,
                             dict(a_kc=False, b_kc=False)))
        sums.append((parts, S, n1 - n0, K, out[n0:n1], kw.get("beta", 0.0)))
    probs = full + [u for q in range(S) for u in split[q]]       # full-length tiles first (whole rounds), then slice by slice
    if len(probs) > _hip.GEMM_MAX_GROUP:
        return problems, []
    return probs, sums


class _DwBatch:
    """Weight-gradient GEMMs of one autograd node, deferred and issued as ONE grouped launch: the last, partially
    filled round of workgroups of each problem is filled by the next problem's tiles (dW grids are small: the
    proj dW of interm_1b is 576 tiles on 512 workgroup slots).  Bias gradients (column sums) are issued at once."""

    def __init__(self):
        self.problems, self.sinks, self.keep = [], [], []

    def add(self, dy2d, x2d, W, b, M, N, K, colsum_parts=None):
        """queues dW[N,K] = dy^T . x; returns (index of the weight result, bias result)"""
        sw = _GradSink(W)
        self.problems.append((dy2d, x2d, sw.buf, N, K, M, _ld(dy2d, N), _ld(x2d, K), K,
                              dict(a_kc=False, b_kc=False, beta=sw.beta)))
        self.sinks.append(sw)
        self.keep.append((dy2d, x2d))
        gb = None
        if b is not None:
            sb = _GradSink(b)
            if colsum_parts is not None:      # per-tile-row partial sums written by the GEMM that produced dy2d (fc2's input gradient)
                _hip.colsum(colsum_parts, colsum_parts.shape[0], N, N, sb.buf, beta=sb.beta)
            else:
                _hip.colsum(dy2d, M, N, _ld(dy2d, N), sb.buf, beta=sb.beta)
            gb = sb.done()
        return len(self.sinks) - 1, gb

    def flush(self):
        """launches the group; returns the per-problem values backward must return for the weights"""
        probs, sums = _dw_balance(self.problems)
        for i in range(0, len(probs), _hip.GEMM_MAX_GROUP):     # the whole node as ONE grouped launch
            _hip.gemm_grouped(probs[i:i + _hip.GEMM_MAX_GROUP])
        for parts, S, rows, K, dst, beta in sums:               # the split tiles' partial products, summed in a fixed order
            _hip.batch_sum(parts, S, rows, K, dst, beta=beta)
        out = [s.done() for s in self.sinks]
        self.problems, self.sinks, self.keep = [], [], []
        return out


def _ln_bwd(dy, x, gamma_p, beta_p, mean, rstd, dres):
    sg, sb = _GradSink(gamma_p), _GradSink(beta_p)
    assert sg.beta == sb.beta
    dx = _hip.layernorm_bwd(dy, x, cw(gamma_p), mean, rstd, dres, sg.buf, sb.buf, beta_acc=sg.beta)
    return dx, sg.done(), sb.done()


def _drop_bwd(dy, M, N, p, seed, rowscale, rps):
    if p > 0.0 or rowscale is not None:
        return _hip.dropout_bwd(dy, M, N, p, seed, rowscale, rps)
    return dy


def _drop_bwd_bias(dy, M, N, p, seed, rowscale, rps, b):
    """dropout / DropPath backward of a Linear's output gradient together with that Linear's bias gradient (column sums of
    the result) in ONE pass over dy; returns (dym, what backward must return for b, bias handled?)"""
    if b is None or not (p > 0.0 or rowscale is not None) or N % 8:
        return _drop_bwd(dy, M, N, p, seed, rowscale, rps), None, False
    sb = _GradSink(b)
    dym = _hip.dropout_bwd_colsum(dy, M, N, p, seed, rowscale, rps, sb.buf, beta=sb.beta)
    return dym, sb.done(), True


# ------------------------------------------------------------------------------------------------------
# transformer block   (reference: components/vit_blocks.py:76-81, attention.py:43-87, mlp.py:57-73)
# ------------------------------------------------------------------------------------------------------
class BlockFn(torch.autograd.Function):
    """x2 = x1 + DropPath(Mlp(LN2(x1))),  x1 = x + DropPath(Attn(LN1(x))) -- 7 kernels forward, 15 backward."""

    @staticmethod
    def forward(ctx, x, cfg, n1w, n1b, wqkv, bqkv, wp, bp, n2w, n2b, w1, b1, w2, b2):
        B, L, D = x.shape
        grp = cfg.get("tp_group")
        tp = _tp.group_size(grp)
        H = cfg["heads"] // tp                     # heads / hidden units held by this tensor-parallel rank
        d = D // cfg["heads"]
        M = B * L
        hid = w1.shape[0]
        p_attn, p_proj, p_mlp, p_path = cfg["attn_drop"], cfg["proj_drop"], cfg["mlp_drop"], cfg["drop_path"]
        sa = tp_seed(seeds.next(), _tp.group_rank(grp)) if p_attn > 0 else 0
        sp = seeds.next() if p_proj > 0 else 0
        s1 = tp_seed(seeds.next(), _tp.group_rank(grp)) if p_mlp > 0 else 0
        s2 = seeds.next() if p_mlp > 0 else 0
        dp1 = dp2 = None
        if p_path > 0:
            dp1 = _hip.droppath_scales(B, p_path, seeds.next(), x.device)
            dp2 = _hip.droppath_scales(B, p_path, seeds.next(), x.device)
        x2d = x.reshape(M, D)
        saved = BlockFn._run(x2d, B, L, D, H, d, M, hid, (p_attn, p_proj, p_mlp), (sa, sp, s1, s2), dp1, dp2,
                             (n1w, n1b, wqkv, bqkv, wp, bp, n2w, n2b, w1, b1, w2, b2), grp)
        x2 = saved[-1]
        ctx.meta = (B, L, D, H, d, M, hid, (p_attn, p_proj, p_mlp), (sa, sp, s1, s2), cfg.get("recompute", False))
        ctx.params = (n1w, n1b, wqkv, bqkv, wp, bp, n2w, n2b, w1, b1, w2, b2)
        ctx.grp = grp
        if ctx.meta[-1]:
            ctx.save_for_backward(x2d, dp1, dp2)
        else:
            ctx.save_for_backward(x2d, dp1, dp2, *saved[:-1])
        return x2.view(B, L, D)

    @staticmethod
    def _run(x2d, B, L, D, H, d, M, hid, ps, sds, dp1, dp2, prm, grp=None):
        n1w, n1b, wqkv, bqkv, wp, bp, n2w, n2b, w1, b1, w2, b2 = prm
        p_attn, p_proj, p_mlp = ps
        sa, sp, s1, s2 = sds
        Dl = H * d                                  # = D on one rank, D / tensor_par_size under head-split
        # h1, h2 and o are GEMM A operands: they carry _ld_pad's row pitch (a no-op unless D is a multiple of 4096 -- interm_10b --
        # where rows 16 KiB apart would put every row's k-offset on one memory channel: DESIGN 4.1)
        h1, mean1, rstd1 = _hip.layernorm_fwd(x2d, cw(n1w), cw(n1b), out=_rows(M, D, x2d.device))
        # the q third leaves the GEMM epilogue as q * log2(e)/sqrt(d) (fp32 product, ONE rounding to bf16): the attention
        # kernels' scores are exp2 arguments with exact bf16 x bf16 products, as with the reference's fp32 scaling
        qkv = _linear_fwd(h1, wqkv, bqkv, M, 3 * Dl, D, pad=True, colscale=(Dl, _Q_PRESCALE / math.sqrt(d)))
        o2d, lse = _hip.attn_fwd(qkv, B, L, H, d, p_attn, sa, flags=_hip.ATTN_Q_PRESCALED, out=_rows(M, Dl, x2d.device))
        if grp is None:
            x1 = _linear_fwd(o2d, wp, bp, M, D, Dl, drop_p=p_proj, seed=sp, rowscale=dp1, rows_per_scale=L,
                             residual=x2d, ldr=D)
        else:
            # row-parallel proj: bias + dropout on the partial product (identical masks on all ranks of the group),
            # SUM over the group, then DropPath scale + residual (reference attention.py:81-85, vit_blocks.py:77)
            part = _linear_fwd(o2d, wp, bp, M, D, Dl, drop_p=p_proj, seed=sp)
            _tp.all_reduce_sum(part, grp)
            x1 = _hip.post_reduce(part, M, D, residual=x2d, rowscale=dp1, rows_per_scale=L)
        h2, mean2, rstd2 = _hip.layernorm_fwd(x1, cw(n2w), cw(n2b), out=_rows(M, D, x2d.device))
        # the hidden tensors (gsaved, hm, and dpre in backward) carry _ld_pad's row pitch.  `gsaved` (what the GELU layer
        # saved) holds what the backward multiplies
        # by -- GELU'(pre-activation) x dropout factor, computed in fc1's epilogue where both are in registers (save_dact) --
        # not the pre-activation itself: the fc2 input gradient then has a one-multiply epilogue (4-wave kernel) instead of
        # GELU' + the mask again (reference: autograd of mlp.py:64-65)
        gsaved = _rows(M, hid, x2d.device, _gelu_saved_dtype(p_mlp))      # the factor tensor (int16 q14) or the pre-activation
        hm = _linear_fwd(h2, w1, b1, M, hid, D, pad=True, **_gelu_fwd_kw(gsaved, p_mlp, s1))
        if grp is None:
            x2 = _linear_fwd(hm, w2, b2, M, D, hid, drop_p=p_mlp, seed=s2, rowscale=dp2, rows_per_scale=L,
                             residual=x1, ldr=D)
        else:
            part = _linear_fwd(hm, w2, b2, M, D, hid, drop_p=p_mlp, seed=s2)
            _tp.all_reduce_sum(part, grp)
            x2 = _hip.post_reduce(part, M, D, residual=x1, rowscale=dp2, rows_per_scale=L)
        return h1, mean1, rstd1, qkv, o2d, lse, x1, h2, mean2, rstd2, gsaved, hm, x2

    @staticmethod
    def backward(ctx, dx2):
        B, L, D, H, d, M, hid, ps, sds, recompute = ctx.meta
        p_attn, p_proj, p_mlp = ps
        sa, sp, s1, s2 = sds
        n1w, n1b, wqkv, bqkv, wp, bp, n2w, n2b, w1, b1, w2, b2 = ctx.params
        if recompute:
            x2d, dp1, dp2 = ctx.saved_tensors
            h1, mean1, rstd1, qkv, o2d, lse, x1, h2, mean2, rstd2, gsaved, hm, _ = BlockFn._run(
                x2d, B, L, D, H, d, M, hid, ps, sds, dp1, dp2, ctx.params, ctx.grp)
        else:
            x2d, dp1, dp2, h1, mean1, rstd1, qkv, o2d, lse, x1, h2, mean2, rstd2, gsaved, hm = ctx.saved_tensors
        grp, Dl = ctx.grp, H * d
        dx2 = dx2.reshape(M, D)
        if dx2.dtype != BF or not dx2.is_contiguous():
            dx2 = dx2.contiguous().to(BF)
        # ---- MLP branch   (the four dW GEMMs are queued and issued as one grouped launch at the end)
        dws = _DwBatch()
        dym2, gb2, done2 = _drop_bwd_bias(dx2, M, D, p_mlp, s2, dp2, L, b2)      # fc2's bias gradient rides along
        i2, gb2_ = dws.add(dym2, hm, w2, None if done2 else b2, M, D, hid)
        gb2 = gb2 if done2 else gb2_
        # (fc1's bias gradient = column sums of dpre: the factor-multiply epilogue of this GEMM leaves them per tile row when it can)
        dpre, dpre_sums = _dx(dym2, w2, M, D, hid, pad=True, want_colsum=_FUSE_COLSUM, **_gelu_bwd_kw(gsaved, p_mlp, s1)), None
        if _FUSE_COLSUM:
            dpre, dpre_sums = dpre
        del hm, gsaved, dym2
        i1, gb1 = dws.add(dpre, h2, w1, b1, M, hid, D, colsum_parts=dpre_sums)
        dh2 = _dx(dpre, w1, M, hid, D)
        del dpre, h2
        _tp.all_reduce_sum(dh2, grp)        # column-parallel fc1: the input gradient is a partial sum per rank
        dx1, gn2w, gn2b = _ln_bwd(dh2, x1, n2w, n2b, mean2, rstd2, dx2)
        del dh2, x1
        # ---- attention branch
        dym1, gbp, donep = _drop_bwd_bias(dx1, M, D, p_proj, sp, dp1, L, bp)     # proj's bias gradient rides along
        ip, gbp_ = dws.add(dym1, o2d, wp, None if donep else bp, M, D, Dl)
        gbp = gbp if donep else gbp_
        do = _dx(dym1, wp, M, D, Dl)
        del dym1
        # (dqkv is the gradient with respect to the UNSCALED q, k, v: the qkv GEMM's backward needs no column scale)
        dqkv = _hip.attn_bwd(qkv, o2d, do, lse, B, L, H, d, p_attn, sa, flags=_hip.ATTN_Q_PRESCALED)
        del do, o2d, qkv
        iq, gbqkv = dws.add(dqkv, h1, wqkv, bqkv, M, 3 * Dl, D)
        dh1 = _dx(dqkv, wqkv, M, 3 * Dl, D)
        del dqkv, h1
        _tp.all_reduce_sum(dh1, grp)        # column-parallel qkv
        dx, gn1w, gn1b = _ln_bwd(dh1, x2d, n1w, n1b, mean1, rstd1, dx1)
        gws = dws.flush()
        gw2, gw1, gwp, gwqkv = gws[i2], gws[i1], gws[ip], gws[iq]
        return (dx.view(B, L, D), None, gn1w, gn1b, gwqkv, gbqkv, gwp, gbp, gn2w, gn2b, gw1, gb1, gw2, gb2)


# ------------------------------------------------------------------------------------------------------
# chain of Linear(+GELU) layers with an optional leading LayerNorm
#   head:  LN -> (Linear -> GELU) x dd -> Linear          (res_slimvit.py:104,115-120,294,326)
#   Mlp :  Linear -> GELU -> Dropout -> Linear -> Dropout   (mlp.py:57-73, stand-alone use)
# ------------------------------------------------------------------------------------------------------
class ChainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg, *prm):
        lead = x.shape[:-1]
        D0 = x.shape[-1]
        M = x.numel() // D0
        x2d = x.reshape(M, D0)
        has_ln = cfg["ln"]
        p_mid, p_out = cfg.get("p_mid", 0.0), cfg.get("p_out", 0.0)
        off = 2 if has_ln else 0
        layers = [(prm[off + 2 * i], prm[off + 2 * i + 1]) for i in range((len(prm) - off) // 2)]
        saved = []
        if has_ln:
            h, mean, rstd = _hip.layernorm_fwd(x2d, cw(prm[0]), cw(prm[1]))
            saved += [x2d, mean, rstd]
        else:
            h = x2d
        sds = []
        for i, (W, b) in enumerate(layers):
            N, K = W.shape
            last = i == len(layers) - 1
            if last:
                s = seeds.next() if p_out > 0 else 0
                y = _linear_fwd(h, W, b, M, N, K, drop_p=p_out, seed=s)
                saved.append(h)
            else:
                s = seeds.next() if p_mid > 0 else 0
                gsaved = torch.empty(M, N, dtype=_gelu_saved_dtype(p_mid), device=x.device)
                y = _linear_fwd(h, W, b, M, N, K, **_gelu_fwd_kw(gsaved, p_mid, s))
                saved += [h, gsaved]
            sds.append(s)
            h = y
        ctx.meta = (M, has_ln, p_mid, p_out, sds, lead, D0)
        ctx.prm = prm
        ctx.save_for_backward(*saved)
        return h.view(*lead, h.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        M, has_ln, p_mid, p_out, sds, lead, D0 = ctx.meta
        prm = ctx.prm
        sv = list(ctx.saved_tensors)
        off = 2 if has_ln else 0
        layers = [(prm[off + 2 * i], prm[off + 2 * i + 1]) for i in range((len(prm) - off) // 2)]
        if has_ln:
            x2d, mean, rstd = sv[:3]
            sv = sv[3:]
        nl = len(layers)
        dy = dy.reshape(M, dy.shape[-1])
        if dy.dtype != BF or not dy.is_contiguous():
            dy = dy.contiguous().to(BF)
        g = _drop_bwd(dy, M, dy.shape[-1], p_out, sds[-1], None, 0)
        grads = [None] * (2 * nl)
        # the weight gradients of the wide layers go out as ONE grouped launch at the end (alone, a D x D weight gradient is
        # 144 tiles of 256 x 256: half a round of the chip, so it fell back to the 128-tile kernel at 680 TFLOP/s)
        dws, queued = _DwBatch(), []
        for i in range(nl - 1, -1, -1):
            W, b = layers[i]
            N, K = W.shape
            h_in = sv[2 * i] if i < nl - 1 else sv[2 * (nl - 1)]
            if min(N, K) >= 256:
                idx, grads[2 * i + 1] = dws.add(g, h_in, W, b, M, N, K)
                queued.append((i, idx))
            else:
                grads[2 * i], grads[2 * i + 1] = _dw(g, h_in, W, b, M, N, K)
            if i > 0:
                gsaved_prev = sv[2 * (i - 1) + 1]
                g = _dx(g, W, M, N, K, **_gelu_bwd_kw(gsaved_prev, p_mid, sds[i - 1]))
            elif has_ln or ctx.needs_input_grad[0]:
                g = _dx(g, W, M, N, K)
        if queued:
            gws = dws.flush()
            for i, idx in queued:
                grads[2 * i] = gws[idx]
        out = [None, None]
        if has_ln:
            dx, gg, gb = _ln_bwd(g, x2d, prm[0], prm[1], mean, rstd, None)
            out[0] = dx.view(*lead, D0)
            out += [gg, gb]
        else:
            out[0] = g.view(*lead, D0) if ctx.needs_input_grad[0] else None
        return tuple(out + grads)


class LinearFn(torch.autograd.Function):
    """y = Dropout(x W^T + b) (+ residual); stand-alone Linear (Attention.qkv / .proj used outside a Block)."""

    @staticmethod
    def forward(ctx, x, W, b, p_drop, residual):
        lead = x.shape[:-1]
        K = x.shape[-1]
        N = W.shape[0]
        M = x.numel() // K
        x2d = x.reshape(M, K)
        s = seeds.next() if p_drop > 0 else 0
        r2d = None if residual is None else residual.reshape(M, N)
        y = _linear_fwd(x2d, W, b, M, N, K, drop_p=p_drop, seed=s, residual=r2d, ldr=N)
        ctx.meta = (M, N, K, p_drop, s, lead)
        ctx.prm = (W, b)
        ctx.has_res = residual is not None
        ctx.save_for_backward(x2d)
        return y.view(*lead, N)

    @staticmethod
    def backward(ctx, dy):
        M, N, K, p, s, lead = ctx.meta
        W, b = ctx.prm
        (x2d,) = ctx.saved_tensors
        dy2 = dy.reshape(M, N)
        if dy2.dtype != BF or not dy2.is_contiguous():
            dy2 = dy2.contiguous().to(BF)
        g = _drop_bwd(dy2, M, N, p, s, None, 0)
        gw, gb = _dw(g, x2d, W, b, M, N, K)
        dx = _dx(g, W, M, N, K).view(*lead, K) if ctx.needs_input_grad[0] else None
        return dx, gw, gb, None, (dy if ctx.has_res else None)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        D = x.shape[-1]
        x2d = x.reshape(-1, D)
        y, mean, rstd = _hip.layernorm_fwd(x2d, cw(w), cw(b))
        ctx.prm = (w, b)
        ctx.save_for_backward(x2d, mean, rstd)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2d, mean, rstd = ctx.saved_tensors
        dy2 = dy.reshape(x2d.shape).contiguous().to(BF)
        dx, gg, gb = _ln_bwd(dy2, x2d, ctx.prm[0], ctx.prm[1], mean, rstd, None)
        return dx.view(dy.shape), gg, gb


class AttnCoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v on the packed qkv activation [B, L, 3*H*d] (attention.py:50-78)."""

    @staticmethod
    def forward(ctx, qkv, H, p_drop, tp_rank=0):
        B, L, C3 = qkv.shape
        d = C3 // (3 * H)
        s = tp_seed(seeds.next(), tp_rank) if p_drop > 0 else 0
        q = qkv.contiguous()
        o, lse = _hip.attn_fwd(q, B, L, H, d, p_drop, s)
        ctx.meta = (B, L, H, d, p_drop, s)
        ctx.save_for_backward(q, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        B, L, H, d, p, s = ctx.meta
        q, o, lse = ctx.saved_tensors
        dq = _hip.attn_bwd(q, o, do.contiguous().to(BF), lse, B, L, H, d, p, s)
        return dq, None, None, None


# ------------------------------------------------------------------------------------------------------
# small fp32 GEMM with autograd (parameter-table algebra)
# ------------------------------------------------------------------------------------------------------
class SgemmFn(torch.autograd.Function):
    """C = op(A) op(B), fp32, op = transpose if flagged."""

    @staticmethod
    def forward(ctx, A, B, ta, tb):
        A, B = A.contiguous(), B.contiguous()
        M = A.shape[1] if ta else A.shape[0]
        K = A.shape[0] if ta else A.shape[1]
        N = B.shape[0] if tb else B.shape[1]
        out = torch.empty(M, N, dtype=F32, device=A.device)
        _hip.sgemm(A, B, out, M, N, K, A.shape[1], B.shape[1], N, ta=ta, tb=tb)
        ctx.meta = (ta, tb, M, N, K)
        ctx.save_for_backward(A, B)
        return out

    @staticmethod
    def backward(ctx, dC):
        ta, tb, M, N, K = ctx.meta
        A, B = ctx.saved_tensors
        dC = dC.contiguous()
        dA = dB = None
        if ctx.needs_input_grad[0]:
            dA = torch.empty_like(A)
            if not ta:   # dA[M,K] = dC[M,N] . op(B)^T ; op(B)[K,N]
                _hip.sgemm(dC, B, dA, M, K, N, N, B.shape[1], K, ta=False, tb=not tb)
            else:        # A stored [K,M]: dA[K,M] = op(B)[K,N] . dC^T[N,M]
                _hip.sgemm(B, dC, dA, K, M, N, B.shape[1], N, M, ta=tb, tb=True)
        if ctx.needs_input_grad[1]:
            dB = torch.empty_like(B)
            if not tb:   # B stored [K,N]: dB = op(A)^T[K,M] . dC[M,N]
                _hip.sgemm(A, dC, dB, K, N, M, A.shape[1], N, N, ta=not ta, tb=False)
            else:        # B stored [N,K]: dB[N,K] = dC^T[N,M] . op(A)[M,K]
                _hip.sgemm(dC, A, dB, N, K, M, N, A.shape[1], K, ta=True, tb=ta)
        return dA, dB, None, None


def sgemm(A, B, ta=False, tb=False):
    return SgemmFn.apply(A, B, ta, tb)


# ------------------------------------------------------------------------------------------------------
# embedding front-end: folded patch-embed + variable aggregation + proj + pos/res embedding + dropout
#   (res_slimvit.py:250-284, attention.py:132-183, patch_embed.py:44-52)
# ------------------------------------------------------------------------------------------------------
class TokenTablesFn(torch.autograd.Function):
    """cmat [5 V, D]: per used variable the four transposed patch-embed weight rows and bias + var_embed row (reference
    res_slimvit.py:64-66, 182-201, 251-262) -- ONE launch forward; backward, the transpose accumulates straight into the engine's
    flat fp32 gradient bucket (the per-variable parameters lie there at a uniform pitch) and tells the engine the parameters are
    ready: no per-parameter autograd accumulation (2 V + 1 tiny launches per step in the small configurations).
    Only for engine-managed parameters at a uniform pitch (`token_tables_layout`); the model keeps the ATen path otherwise."""
    @staticmethod
    def forward(ctx, layout, ids_t, V, D, var_embed, *te_params):
        w0, ws, b0, bs = layout["w0"], layout["ws"], layout["b0"], layout["bs"]
        ctx.layout, ctx.ids_t, ctx.V, ctx.D = layout, ids_t, V, D
        ctx.params = (var_embed,) + tuple(te_params)
        if "pending" in layout:
            layout["pending"][0] += 1                     # an instance of this step is outstanding until its backward has run
        return _hip.tables_gather(w0.data, ws, b0.data, bs, var_embed.data.reshape(-1, D), ids_t, V, D)

    @staticmethod
    def backward(ctx, dcmat):
        lay = ctx.layout
        dcmat = dcmat.contiguous()
        if dcmat.dtype != F32:
            dcmat = dcmat.float()
        var_embed = ctx.params[0]
        _hip.tables_scatter(dcmat, lay["w0"].grad, lay["ws"], lay["b0"].grad, lay["bs"], var_embed.grad.view(-1, ctx.D), ctx.ids_t,
                            ctx.V, ctx.D)
        for p in ctx.params:                              # the engine counts these parameters as arrived (no AccumulateGrad runs)
            p._o2_engine.grad_ready(p)
        if "pending" in lay:
            lay["pending"][0] = max(0, lay["pending"][0] - 1)
        return (None,) * (5 + len(ctx.params) - 1)


def token_tables_layout(token_embeds, var_embed):
    """{w0, ws, b0, bs} if every per-variable patch-embed weight / bias (and their gradient views) lies at one uniform pitch in an
    engine's flat buffers and everything requires a gradient; None otherwise (plain module, frozen parameters, other engines)"""
    ws_ = [te.proj.weight for te in token_embeds]
    bs_ = [te.proj.bias for te in token_embeds]
    ps = ws_ + bs_ + [var_embed]
    if any(getattr(p, "_o2_engine", None) is None or p.grad is None or not p.requires_grad or p.dtype != F32 or not p.is_cuda
           for p in ps):
        return None
    if len({id(p._o2_engine) for p in ps}) != 1 or not var_embed.grad.is_contiguous():
        return None
    if len(ws_) == 1:
        return {"w0": ws_[0], "ws": 0, "b0": bs_[0], "bs": 0}
    pitch = lambda ts, attr: {(getattr(b, attr).data_ptr() - getattr(a, attr).data_ptr()) for a, b in zip(ts[:-1], ts[1:])}
    out = {}
    for key, ts in (("w", ws_), ("b", bs_)):
        pd, pg = pitch(ts, "data"), pitch(ts, "grad")
        if len(pd) != 1 or pd != pg:
            return None
        step = pd.pop()
        if step <= 0 or step % 16:
            return None
        out[key + "0"], out[key + "s"] = ts[0], step // 4
    return out


class PosResFn(torch.autograd.Function):
    """the [L, D] fp32 table a step adds to its tokens: pos_embed re-gridded to the run's token grid (bicubic, only when the
    heights differ -- components/pos_embed.py:103-138) + the resolution embedding Linear(1, D)(res) (res_slimvit.py:277-281);
    one HIP launch forward, the re-grid's transpose + two column sums backward"""
    @staticmethod
    def forward(ctx, pos_embed, sw, sb, res, oh, ow, nh, nw):
        pe = pos_embed.reshape(oh * ow, pos_embed.shape[-1])
        ctx.meta = (float(res), oh, ow, nh, nw, pos_embed.shape)
        return _hip.posembed_fwd(pe.contiguous(), sw.reshape(-1).contiguous(), sb.contiguous(), float(res), oh, ow, nh, nw)

    @staticmethod
    def backward(ctx, dout):
        res, oh, ow, nh, nw, shape = ctx.meta
        dout = dout.contiguous()
        if dout.dtype != F32:
            dout = dout.float()
        dpe = dsw = dsb = None
        if ctx.needs_input_grad[0]:
            dpe = _hip.posembed_bwd(dout, oh, ow, nh, nw).view(shape)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            L, D = dout.shape
            dsb = torch.empty(D, dtype=F32, device=dout.device)
            _hip.colsum(dout, L, D, D, dsb)
            dsw = (dsb * res).view(D, 1)
        return dpe, dsw, dsb, None, None, None, None, None


class EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xgrid, stab, gtab, posres, wp, bp, H, p_drop, grp=None):
        """H = heads held by this rank; under head-split tensor parallelism wp is [D, D / tp] and the proj output is
        summed over `grp` before the position / resolution embedding and the dropout are applied."""
        B, V, h, w = xgrid.shape
        D, Dl = wp.shape
        L = (h // 2) * (w // 2)
        M = B * L
        xg = xgrid.contiguous()
        if xg.dtype != F32:
            xg = xg.float()
        st, gt = stab.contiguous(), gtab.contiguous()
        z, attw = _hip.varagg_fwd(xg, st, gt, H, Dl)
        pr16 = _hip.cast_to_bf16(posres.contiguous())
        s = seeds.next() if p_drop > 0 else 0
        if grp is None:
            tok = _linear_fwd(z, wp, bp, M, D, Dl, drop_p=p_drop, seed=s, residual=pr16, ldr=D, res_mod=L,
                              res_first=True)
        else:
            tok = _linear_fwd(z, wp, bp, M, D, Dl)          # row-parallel proj (attention.py:176-181)
            _tp.all_reduce_sum(tok, grp)
            _hip.post_reduce(tok, M, D, addend=pr16, res_mod=L, drop_p=p_drop, seed=s)
        ctx.meta = (B, V, h, w, H, D, Dl, L, M, p_drop, s)
        ctx.prm = (wp, bp)
        ctx.save_for_backward(xg, gt, attw, z)
        return tok.view(B, L, D)

    @staticmethod
    def backward(ctx, dtok):
        B, V, h, w, H, D, Dl, L, M, p, s = ctx.meta
        wp, bp = ctx.prm
        xg, gt, attw, z = ctx.saved_tensors
        d2 = dtok.reshape(M, D)
        if d2.dtype != BF or not d2.is_contiguous():
            d2 = d2.contiguous().to(BF)
        g = _drop_bwd(d2, M, D, p, s, None, 0)
        dposres = None
        if ctx.needs_input_grad[3]:
            dposres = torch.empty(L, D, dtype=F32, device=g.device)
            _hip.batch_sum(g, B, L, D, dposres)
        gwp, gbp = _dw(g, z, wp, bp, M, D, Dl)
        dz = _dx(g, wp, M, D, Dl)
        dstab, dgtab = _hip.varagg_bwd(xg, gt, attw, dz, H, Dl)
        return None, dstab, dgtab, dposres, gwp, gbp, None, None, None


# ------------------------------------------------------------------------------------------------------
# hi-res tail
# ------------------------------------------------------------------------------------------------------
class UnpatchifyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, C, h, w, p, s):
        B = t.shape[0]
        ctx.meta = (B, C, h, w, p, s)
        return _hip.unpatchify_fwd(t.contiguous(), B, C, h, w, p, s)

    @staticmethod
    def backward(ctx, dimg):
        B, C, h, w, p, s = ctx.meta
        return _hip.unpatchify_bwd(dimg.contiguous(), B, C, h, w, p, s), None, None, None, None, None


class Conv3x3Fn(torch.autograd.Function):
    """3x3 conv (pad 1) on fp32 NCHW with optional channel gather, GELU+PixelShuffle epilogue, image addend."""

    @staticmethod
    def forward(ctx, x, weight, bias, chan_idx, mode, r, addend):
        x = x.contiguous()
        if x.dtype != F32:
            x = x.float()
        wd, bd = weight.detach().contiguous(), bias.detach().contiguous()
        out, pre = _hip.conv3x3_fwd(x, chan_idx, wd, bd, mode, r, None if addend is None else addend.contiguous())
        ctx.meta = (mode, r, None if addend is None else tuple(addend.shape))
        ctx.idx = chan_idx
        ctx.save_for_backward(x, wd, pre if pre is not None else x.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, dout):
        mode, r, ashape = ctx.meta
        x, wd, pre = ctx.saved_tensors
        dout = dout.contiguous()
        din, dw, db = _hip.conv3x3_bwd(dout, x, ctx.idx, wd, pre if mode == 1 else None, ctx.needs_input_grad[0],
                                       mode, r)
        dadd = None
        if ashape is not None and ctx.needs_input_grad[6]:
            if ashape == tuple(dout.shape):
                dadd = dout
            else:
                dadd = dout.new_zeros(ashape)
                dadd[:, :, : dout.shape[2], : dout.shape[3]] = dout
        return din, dw, db, None, None, None, dadd


class ClampChannelFn(torch.autograd.Function):
    """in-place clamp of one channel at 0 (examples/intermediate_downscaling.py:267-272)."""

    @staticmethod
    def forward(ctx, img, chan):
        ctx.chan = chan
        ctx.mark_dirty(img)
        _hip.clamp_channel_(img, chan)
        ctx.save_for_backward(img)
        return img

    @staticmethod
    def backward(ctx, g):
        (img,) = ctx.saved_tensors
        g = g.contiguous().clone()
        _hip.clamp_channel_bwd_(img, g, ctx.chan)
        return g, None


class LossFn(torch.autograd.Function):
    """fused mse / bayesian_tv (+lat / variable weights). Returns [C+1] (per-channel means, aggregate)."""

    @staticmethod
    def forward(ctx, pred, target, lat_w, chan_w, kind):
        pred = pred.contiguous()
        if pred.dtype != F32:
            pred = pred.float()
        target = target.contiguous()
        out = _hip.loss_fwd(pred, target, lat_w, chan_w, kind)
        ctx.kind = kind
        ctx.save_for_backward(pred, target, lat_w if lat_w is not None else pred.new_empty(0),
                              chan_w if chan_w is not None else pred.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, g):
        pred, target, lat_w, chan_w = ctx.saved_tensors
        gs = g[-1:].contiguous().float()          # gradient flows through the aggregate entry only
        dp = _hip.loss_bwd(pred, target, lat_w if lat_w.numel() else None, chan_w if chan_w.numel() else None, gs,
                           ctx.kind)
        return dp, None, None, None, None
