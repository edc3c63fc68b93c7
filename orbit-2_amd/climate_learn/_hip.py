"""ctypes binding of liborbit2_hip.so (C ABI in include/orbit2_hip.h).

This is the ONLY compute backend of the package: there is no CPU or eager fallback.  Every wrapper
checks that its operands are contiguous device tensors of the expected dtype and passes raw device
pointers plus torch's current HIP stream across the ABI.  A missing library or a non-zero return
code raises immediately."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# $ORBIT2_HIP_LIB: another build of the same ABI (A/B timing of kernel variants on one box; tools/ab_build.sh)
LIB_PATH = os.environ.get("ORBIT2_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "liborbit2_hip.so")
_lib = None


class HipBackendError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
        ("a_kc", C.c_int), ("b_kc", C.c_int),
        ("bias", C.c_void_p),
        ("act", C.c_int),
        ("save_pre", C.c_void_p),
        ("dgelu_pre", C.c_void_p),
        ("drop_p", C.c_float),
        ("seed", C.c_uint64),
        ("rowscale", C.c_void_p),
        ("rows_per_scale", C.c_int),
        ("residual", C.c_void_p),
        ("ldr", C.c_int), ("res_mod", C.c_int), ("res_first", C.c_int),
        ("out_fp32", C.c_int),
        ("beta", C.c_float),
        ("tile_hint", C.c_int),
        ("colscale_n", C.c_int),
        ("colscale", C.c_float),
        ("save_dact", C.c_void_p),
        ("mul", C.c_void_p),
        ("colsum_ws", C.c_void_p),
    ]


def lib():
    """Load the shared library once; fail loudly if it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipBackendError(
                "liborbit2_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). This package has no CPU fallback." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        if _lib.orbit2_abi_version() != 6:
            raise HipBackendError("liborbit2_hip.so ABI version mismatch")
    return _lib


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk(rc: int, name: str):
    if rc != 0:
        raise HipBackendError("%s failed with code %d (-1 bad argument, -2 launch error, -3 unsupported)" % (name, rc))


def _dev(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise HipBackendError("%s must be a GPU tensor (HIP backend has no CPU path)" % name)
    if t.dtype != dtype:
        raise HipBackendError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise HipBackendError("%s must be contiguous" % name)
    return t


BF, F32 = torch.bfloat16, torch.float32


def _dev_rows(t: torch.Tensor, dtype, name: str):
    """a GEMM operand / epilogue tensor handed over with an explicit leading dimension: rows contiguous, any row pitch"""
    if t.dim() == 2 and t.stride(1) == 1 and t.is_cuda and t.dtype == dtype:
        return t
    return _dev(t, dtype, name)


class KernelTimer:
    """Optional live timing of individual launches with HIP events recorded on the launch stream
    (bench.py uses it for the roofline of the dominant kernel).  Off unless `_hip.timer` is set."""

    def __init__(self):
        self.records = {}      # name -> [ (work, ev0, ev1) ]

    def span(self, name, work, nbytes=0.0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.setdefault(name, []).append((work, e0, e1, nbytes))
        return e0, e1

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [r[1].elapsed_time(r[2]) for r in recs]
            out[name] = {"launches": len(recs), "work": float(sum(r[0] for r in recs)), "ms": float(sum(ms)),
                         "bytes": float(sum(r[3] for r in recs)) / max(1, len(recs))}
        return out


timer: Optional[KernelTimer] = None


# ------------------------------------------------------------------------------------------------
def _gemm_fill(a, A, B, out, M, N, K, lda, ldb, ldc, a_kc=True, b_kc=True, bias=None, act=0, save_pre=None,
               dgelu_pre=None, drop_p=0.0, seed=0, rowscale=None, rows_per_scale=0, residual=None, ldr=0, res_mod=0,
               res_first=False, beta=0.0, tile=0, colscale=None, save_dact=None, mul=None):
    for t, nm in ((A, "A"), (B, "B")):
        _dev_rows(t, BF, nm)
    if out.dtype not in (BF, F32) or not out.is_cuda:
        raise HipBackendError("gemm out must be a bf16/fp32 GPU tensor")
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldb, a.ldc = M, N, K, lda, ldb, ldc
    a.a_kc, a.b_kc = int(a_kc), int(b_kc)
    a.bias = None if bias is None else _dev(bias, BF, "bias").data_ptr()
    a.act = act
    a.save_pre = None if save_pre is None else _dev_rows(save_pre, BF, "save_pre").data_ptr()      # row pitch = ldc
    a.dgelu_pre = None if dgelu_pre is None else _dev_rows(dgelu_pre, BF, "dgelu_pre").data_ptr()  # row pitch = ldc
    a.drop_p, a.seed = float(drop_p), int(seed) & 0xFFFFFFFFFFFFFFFF
    a.rowscale = None if rowscale is None else _dev(rowscale, F32, "rowscale").data_ptr()
    a.rows_per_scale = rows_per_scale
    a.residual = None if residual is None else _dev(residual, BF, "residual").data_ptr()
    a.ldr, a.res_mod, a.res_first = ldr, res_mod, int(res_first)
    a.out_fp32 = int(out.dtype == F32)
    a.beta = float(beta)
    a.tile_hint = int(tile)
    a.colscale_n, a.colscale = (0, 1.0) if colscale is None else (int(colscale[0]), float(colscale[1]))
    a.save_dact = None if save_dact is None else _dev_rows(save_dact, torch.int16, "save_dact").data_ptr()    # int16 q14, row pitch = ldc
    a.mul = None if mul is None else _dev_rows(mul, torch.int16, "mul").data_ptr()                   # int16 q14, row pitch = ldc
    a.colsum_ws = None
    return 2.0 * M * N * K, 2.0 * (M * K + N * K) + M * N * (4.0 if out.dtype == F32 else 2.0)


def gemm(A, B, out, M, N, K, lda, ldb, ldc, want_colsum=False, **kw):
    """out[M,N] = epilogue(A x B); see include/orbit2_hip.h:orbit2_gemm_bf16.
    want_colsum: returns (out, parts) -- parts = fp32 [M / 256, N] per-tile-row column sums of the stored output when this call
    can fuse them (orbit2_gemm_bf16_colsum_rows), else None: the caller then runs `colsum` on `out` itself."""
    a = GemmArgs()
    flops, nbytes = _gemm_fill(a, A, B, out, M, N, K, lda, ldb, ldc, **kw)
    parts = None
    if want_colsum:
        rows = lib().orbit2_gemm_bf16_colsum_rows(C.byref(a))
        if rows > 0:
            parts = torch.empty(rows, N, dtype=F32, device=out.device)
            a.colsum_ws = parts.data_ptr()
    if timer is not None:
        e0, e1 = timer.span("gemm_bf16", flops, nbytes)
        e0.record()
        _chk(lib().orbit2_gemm_bf16(C.byref(a), _stream()), "orbit2_gemm_bf16")
        e1.record()
    else:
        _chk(lib().orbit2_gemm_bf16(C.byref(a), _stream()), "orbit2_gemm_bf16")
    return (out, parts) if want_colsum else out


GEMM_MAX_GROUP = 12


def gemm_grouped(problems):
    """problems: list of (A, B, out, M, N, K, lda, ldb, ldc, kwargs) sharing one operand form; one launch
    (include/orbit2_hip.h:orbit2_gemm_bf16_grouped)."""
    n = len(problems)
    if not 0 < n <= GEMM_MAX_GROUP:
        raise HipBackendError("gemm_grouped takes 1..%d problems" % GEMM_MAX_GROUP)
    arr = (GemmArgs * n)()
    flops = nbytes = 0.0
    for i, (A, B, out, M, N, K, lda, ldb, ldc, kw) in enumerate(problems):
        f, b = _gemm_fill(arr[i], A, B, out, M, N, K, lda, ldb, ldc, **kw)
        flops += f
        nbytes += b
    if timer is not None:
        e0, e1 = timer.span("gemm_bf16", flops, nbytes)
        e0.record()
        _chk(lib().orbit2_gemm_bf16_grouped(arr, n, _stream()), "orbit2_gemm_bf16_grouped")
        e1.record()
        return
    _chk(lib().orbit2_gemm_bf16_grouped(arr, n, _stream()), "orbit2_gemm_bf16_grouped")


def sgemm(A, B, out, M, N, K, lda, ldb, ldc, ta=False, tb=False, alpha=1.0, beta=0.0):
    for t, nm in ((A, "A"), (B, "B"), (out, "C")):
        _dev(t, F32, nm)
    L = lib()
    L.orbit2_sgemm_f32_ws_floats.restype = C.c_int64
    n = int(L.orbit2_sgemm_f32_ws_floats(M, N, K))
    ws = torch.empty(n, dtype=F32, device=out.device) if n else None
    _chk(L.orbit2_sgemm_f32_ws(_p(A), _p(B), _p(out), M, N, K, lda, ldb, ldc, int(ta), int(tb), C.c_float(alpha),
                               C.c_float(beta), _p(ws), C.c_int64(n), _stream()), "orbit2_sgemm_f32_ws")
    return out


def layernorm_fwd(x, gamma, beta, eps=1e-5, out=None):
    """out: optional [rows, D] bf16 destination with any row pitch (a view of a padded buffer: orbit2_layernorm_fwd_ld)"""
    _dev(x, BF, "x"); _dev(gamma, BF, "gamma"); _dev(beta, BF, "beta")
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x) if out is None else _dev_rows(out, BF, "out")
    ldy = D if out is None else y.stride(0)
    mean = torch.empty(rows, dtype=F32, device=x.device)
    rstd = torch.empty(rows, dtype=F32, device=x.device)
    _chk(lib().orbit2_layernorm_fwd_ld(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, D, ldy, C.c_float(eps),
                                       _stream()), "orbit2_layernorm_fwd_ld")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, beta_acc=0.0):
    _dev(dy, BF, "dy"); _dev(x, BF, "x"); _dev(gamma, BF, "gamma")
    D = x.shape[-1]
    rows = x.numel() // D
    dx = torch.empty_like(x)
    n = lib().orbit2_layernorm_bwd_ws_floats(rows, D)
    ws = torch.empty(n, dtype=F32, device=x.device)
    fp32 = int(dgamma.dtype == F32)
    _chk(lib().orbit2_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dgamma),
                                    _p(dbeta), fp32, C.c_float(beta_acc), _p(ws), n, rows, D, _stream()),
         "orbit2_layernorm_bwd")
    return dx


ATTN_4WAVES, ATTN_SPLIT_DKV, ATTN_Q_PRESCALED, ATTN_NO_W4 = 1, 2, 4, 8      # include/orbit2_hip.h: kernel-variant flags of the *_ex attention entries (A/B, tests)


def probe_read(buf, blocks, inflight, sink):
    """diagnostic calibration stream (include/orbit2_hip.h: orbit2_probe_read)"""
    _chk(lib().orbit2_probe_read(_p(buf), C.c_int64(buf.numel() * buf.element_size()), int(blocks), int(inflight), _p(sink),
                                 _stream()), "orbit2_probe_read")


def mall_calibration():
    """the calibration streams of the memory-side latency probe, in a fixed order that tools/summarize_prof.py `mall` decodes
    from the dispatch order of probe_read_kernel: Infinity-Cache-resident buffer (96 MB; first sweep = fill, not counted) at
    low / high load, then a 4 GB buffer (HBM) at low / high load"""
    sink = torch.zeros(1, dtype=F32, device="cuda")
    small = torch.ones(24 << 20, dtype=F32, device="cuda")          # 96 MB: beyond the 32 MB of L2, inside the Infinity Cache
    probe_read(small, 2048, 8, sink)                                 # fill
    for _ in range(6):
        probe_read(small, 64, 1, sink)                               # Infinity-Cache hits, lightly loaded
    for _ in range(6):
        probe_read(small, 2048, 8, sink)                             # ... under a saturating stream
    del small
    big = torch.ones(1 << 30, dtype=F32, device="cuda")             # 4 GB: every sweep reads HBM
    for _ in range(2):
        probe_read(big, 64, 1, sink)
    for _ in range(2):
        probe_read(big, 2048, 8, sink)
    del big
    torch.cuda.synchronize()


def attn_fwd(qkv, B, L, H, d, drop_p=0.0, seed=0, flags=0, out=None):
    """out: optional [B * L, H * d] bf16 destination with any token-row pitch (orbit2_attn_fwd_ld); default [B, L, H * d]"""
    _dev_rows(qkv, BF, "qkv")
    ldq = qkv.stride(0) if qkv.dim() == 2 else 3 * H * d       # [B * L, 3 * H * d] with a token-row pitch, or contiguous
    if out is None:
        out, ldo = torch.empty(B, L, H * d, dtype=BF, device=qkv.device), H * d
    else:
        out = _dev_rows(out, BF, "out")
        ldo = out.stride(0)
    lse = torch.empty(B, H, L, dtype=F32, device=qkv.device)
    if timer is not None:
        # algorithmic bytes: qkv read once, out + lse written once
        e0, e1 = timer.span("attn_fwd", 4.0 * B * H * L * L * d, 2.0 * 4 * B * L * H * d + 4.0 * B * H * L)
        e0.record()
    _chk(lib().orbit2_attn_fwd_ld(_p(qkv), _p(out), _p(lse), B, L, H, d, C.c_float(drop_p), C.c_uint64(seed), int(flags),
                                  int(ldq), int(ldo), _stream()), "orbit2_attn_fwd_ld")
    if timer is not None:
        e1.record()
    return out, lse


def attn_bwd(qkv, out, dout, lse, B, L, H, d, drop_p=0.0, seed=0, flags=0):
    _dev_rows(qkv, BF, "qkv"); _dev_rows(out, BF, "out"); _dev(dout, BF, "dout"); _dev(lse, F32, "lse")
    ldo = out.stride(0) if out.dim() == 2 else H * d           # [B * L, H * d] with a token-row pitch, or contiguous [B, L, H * d]
    ldq = qkv.stride(0) if qkv.dim() == 2 else 3 * H * d
    if ldq != 3 * H * d:                                        # dqkv carries qkv's pitch (it is a GEMM operand too)
        dqkv = torch.empty(B * L, ldq, dtype=BF, device=qkv.device)[:, :3 * H * d]
    else:
        dqkv = torch.empty_like(qkv)
    lib().orbit2_attn_bwd_ws_floats.restype = C.c_int64
    delta = torch.empty(int(lib().orbit2_attn_bwd_ws_floats(B, L, H)), dtype=F32, device=qkv.device)
    if timer is not None:
        # algorithmic: 2x the forward's FLOPs (recompute not credited); qkv, out, dout read once, dqkv written once
        e0, e1 = timer.span("attn_bwd", 8.0 * B * H * L * L * d, 2.0 * 8 * B * L * H * d + 8.0 * B * H * L)
        e0.record()
    _chk(lib().orbit2_attn_bwd_ld(_p(qkv), _p(out), _p(dout), _p(lse), _p(delta), _p(dqkv), B, L, H, d,
                                  C.c_float(drop_p), C.c_uint64(seed), int(flags), int(ldq), int(ldo), _stream()), "orbit2_attn_bwd_ld")
    if timer is not None:
        e1.record()
    return dqkv


def varagg_fwd(x, stab, gtab, H, D):
    _dev(x, F32, "x"); _dev(stab, F32, "stab"); _dev(gtab, F32, "gtab")
    B, V, h, w = x.shape
    ntok = B * (h // 2) * (w // 2)
    z = torch.empty(ntok, D, dtype=BF, device=x.device)
    attw = torch.empty(ntok, H, V, dtype=F32, device=x.device)
    _chk(lib().orbit2_varagg_fwd(_p(x), _p(stab), _p(gtab), _p(z), _p(attw), B, V, h, w, H, D, _stream()),
         "orbit2_varagg_fwd")
    return z, attw


def _ws(query, args, device):
    """fp32 workspace of a two-stage (slab + fixed-order combine) reduction: `query` is the entry's *_ws_floats function"""
    query.restype = C.c_int64
    n = int(query(*args))
    if n <= 0:
        raise HipBackendError("workspace query failed for %r" % (args,))
    return torch.empty(n, dtype=F32, device=device)


# set once a gradient was accumulated with atomics (summation order not fixed): dist/tp.py ReplicaGuard then exchanges the
# tensor-parallel replicas' gradients instead of relying on their bitwise agreement
atomics_in_grad_path = False


def varagg_bwd(x, gtab, attw, dz, H, D):
    global atomics_in_grad_path
    _dev(x, F32, "x"); _dev(gtab, F32, "gtab"); _dev(attw, F32, "attw"); _dev(dz, BF, "dz")
    B, V, h, w = x.shape
    if not atomics_in_grad_path and not lib().orbit2_varagg_bwd_is_fixed_order(B, V, h, w, H, D):
        atomics_in_grad_path = True
    dstab = torch.zeros(H, V, 5, dtype=F32, device=x.device)
    dgtab = torch.zeros(V, 5, D, dtype=F32, device=x.device)
    ws = _ws(lib().orbit2_varagg_bwd_ws_floats, (B, V, h, w, H, D), x.device)
    _chk(lib().orbit2_varagg_bwd(_p(x), _p(gtab), _p(attw), _p(dz), _p(dstab), _p(dgtab), B, V, h, w, H, D, _p(ws), _stream()),
         "orbit2_varagg_bwd")
    return dstab, dgtab


def tables_gather(w0, w_stride, b0, b_stride, var_embed, ids, V, D):
    """cmat [5 V, D] fp32 from the per-variable patch-embed parameters laid out at a uniform pitch (w0 / b0 = parameter 0)"""
    _dev(w0, F32, "token_embeds.0.proj.weight"); _dev(b0, F32, "token_embeds.0.proj.bias"); _dev(var_embed, F32, "var_embed")
    _dev(ids, torch.int32, "ids")
    cmat = torch.empty(5 * V, D, dtype=F32, device=w0.device)
    _chk(lib().orbit2_tables_gather(_p(w0), C.c_int64(w_stride), _p(b0), C.c_int64(b_stride), _p(var_embed), _p(ids), _p(cmat),
                                    V, D, _stream()), "orbit2_tables_gather")
    return cmat


def tables_scatter(dcmat, gw0, w_stride, gb0, b_stride, gvar_embed, ids, V, D):
    """accumulates the rows' gradient into the parameters' gradient buffers (same pitches)"""
    _dev(dcmat, F32, "dcmat"); _dev(gw0, F32, "dW"); _dev(gb0, F32, "db"); _dev(gvar_embed, F32, "dvar_embed")
    _chk(lib().orbit2_tables_scatter(_p(dcmat), _p(gw0), C.c_int64(w_stride), _p(gb0), C.c_int64(b_stride), _p(gvar_embed),
                                     _p(ids), V, D, _stream()), "orbit2_tables_scatter")


def dropout_bwd(dy, M, N, drop_p, seed, rowscale=None, rows_per_scale=0, out=None):
    _dev(dy, BF, "dy")
    out = torch.empty_like(dy) if out is None else out
    _chk(lib().orbit2_dropout_bwd(_p(dy), _p(out), M, N, C.c_float(drop_p), C.c_uint64(seed), _p(rowscale),
                                  rows_per_scale, _stream()), "orbit2_dropout_bwd")
    return out


def dropout_bwd_colsum(dy, M, N, drop_p, seed, rowscale, rows_per_scale, colsum_out, beta=0.0):
    """dym = dy * dropmask * rowscale and colsum_out[n] (+)= sum_m dym[m][n] in one pass"""
    _dev(dy, BF, "dy")
    out = torch.empty_like(dy)
    n = lib().orbit2_colsum_ws_floats(M, N)
    ws = torch.empty(n, dtype=F32, device=dy.device)
    _chk(lib().orbit2_dropout_bwd_colsum(_p(dy), _p(out), M, N, C.c_float(drop_p), C.c_uint64(seed), _p(rowscale),
                                         rows_per_scale, _p(colsum_out), int(colsum_out.dtype == F32), C.c_float(beta), _p(ws),
                                         n, _stream()), "orbit2_dropout_bwd_colsum")
    return out


def post_reduce(x, M, N, addend=None, res_mod=0, residual=None, drop_p=0.0, seed=0, rowscale=None, rows_per_scale=0,
                out=None):
    """y = residual + rowscale * dropout(x + addend[m % res_mod]); in place on x unless `out` is given"""
    _dev(x, BF, "x")
    for t, nm in ((addend, "addend"), (residual, "residual")):
        if t is not None:
            _dev(t, BF, nm)
    out = x if out is None else out
    _chk(lib().orbit2_post_reduce(_p(x), _p(addend), res_mod, _p(residual), _p(out), M, N, C.c_float(drop_p),
                                  C.c_uint64(seed), _p(rowscale), rows_per_scale, _stream()), "orbit2_post_reduce")
    return out


def colsum(x, M, N, ldx, out, beta=0.0):
    if x.dtype not in (BF, F32):
        raise HipBackendError("colsum input must be bf16/fp32")
    n = lib().orbit2_colsum_ws_floats(M, N)
    ws = torch.empty(n, dtype=F32, device=x.device)
    _chk(lib().orbit2_colsum(_p(x), int(x.dtype == F32), M, N, ldx, _p(out), int(out.dtype == F32), C.c_float(beta),
                             _p(ws), n, _stream()), "orbit2_colsum")
    return out


def batch_sum(x, B, rows, N, out, beta=0.0):
    _dev(x, BF, "x")
    _chk(lib().orbit2_batch_sum(_p(x), _p(out), B, rows, N, int(out.dtype == F32), C.c_float(beta), _stream()),
         "orbit2_batch_sum")
    return out


def cast_to_bf16(src, dst=None):
    _dev(src, F32, "src")
    dst = torch.empty(src.shape, dtype=BF, device=src.device) if dst is None else dst
    _chk(lib().orbit2_cast_f32_to_bf16(_p(src), _p(dst), C.c_int64(src.numel()), _stream()), "orbit2_cast_f32_to_bf16")
    return dst


def cast_to_f32(src, dst=None):
    _dev(src, BF, "src")
    dst = torch.empty(src.shape, dtype=F32, device=src.device) if dst is None else dst
    _chk(lib().orbit2_cast_bf16_to_f32(_p(src), _p(dst), C.c_int64(src.numel()), _stream()), "orbit2_cast_bf16_to_f32")
    return dst


def add_rowvec(a, vec, rows, N):
    _dev(a, BF, "a"); _dev(vec, BF, "vec")
    y = torch.empty_like(a)
    _chk(lib().orbit2_add_rowvec(_p(a), _p(vec), _p(y), rows, N, _stream()), "orbit2_add_rowvec")
    return y


def posembed_fwd(pe, sw, sb, res, oh, ow, nh, nw):
    """[nh*nw, D] fp32 = bicubic re-grid of the [oh*ow, D] table (identity when oh == nh) + sw * res + sb"""
    _dev(pe, F32, "pos_embed")
    D = pe.shape[-1]
    if sw is not None:
        _dev(sw, F32, "spatial_embed.weight"); _dev(sb, F32, "spatial_embed.bias")
    out = torch.empty(nh * nw, D, dtype=F32, device=pe.device)
    _chk(lib().orbit2_posembed_fwd(_p(pe), _p(sw), _p(sb), C.c_float(res), _p(out), oh, ow, nh, nw, D, _stream()),
         "orbit2_posembed_fwd")
    return out


def posembed_bwd(dout, oh, ow, nh, nw):
    _dev(dout, F32, "dposres")
    D = dout.shape[-1]
    dpe = torch.empty(oh * ow, D, dtype=F32, device=dout.device)
    _chk(lib().orbit2_posembed_bwd(_p(dout), _p(dpe), oh, ow, nh, nw, D, _stream()), "orbit2_posembed_bwd")
    return dpe


def unpatchify_fwd(t, B, Cc, h, w, p, s):
    _dev(t, BF, "t")
    img = torch.empty(B, Cc, h * s, w * s, dtype=F32, device=t.device)
    _chk(lib().orbit2_unpatchify_fwd(_p(t), _p(img), B, Cc, h, w, p, s, _stream()), "orbit2_unpatchify_fwd")
    return img


def unpatchify_bwd(dimg, B, Cc, h, w, p, s):
    _dev(dimg, F32, "dimg")
    L = h * w // (p * p)
    dt = torch.empty(B, L, Cc * (s * p) ** 2, dtype=BF, device=dimg.device)
    _chk(lib().orbit2_unpatchify_bwd(_p(dimg), _p(dt), B, Cc, h, w, p, s, _stream()), "orbit2_unpatchify_bwd")
    return dt


def conv3x3_fwd(x, chan_idx, weight, bias, mode=0, r=1, addend=None):
    _dev(x, F32, "in"); _dev(weight, F32, "weight"); _dev(bias, F32, "bias")
    B, ctot, H, W = x.shape
    Cout, Cin = weight.shape[0], weight.shape[1]
    pre = None
    if mode == 0:
        out = torch.empty(B, Cout, H, W, dtype=F32, device=x.device)
    else:
        out = torch.empty(B, Cout // (r * r), H * r, W * r, dtype=F32, device=x.device)
        pre = torch.empty(B, Cout, H, W, dtype=F32, device=x.device)
    Ha = Wa = 0
    if addend is not None:
        _dev(addend, F32, "addend")
        Ha, Wa = addend.shape[2], addend.shape[3]
    _chk(lib().orbit2_conv3x3_fwd(_p(x), _p(chan_idx), ctot, _p(weight), _p(bias), _p(out), _p(pre), _p(addend), Ha, Wa,
                                  B, Cin, Cout, H, W, mode, r, _stream()), "orbit2_conv3x3_fwd")
    return out, pre


def conv3x3_bwd(dout, x, chan_idx, weight, pre, need_din, mode=0, r=1):
    _dev(dout, F32, "dout"); _dev(x, F32, "in"); _dev(weight, F32, "weight")
    B, ctot, H, W = x.shape
    Cout, Cin = weight.shape[0], weight.shape[1]
    din = torch.empty(B, Cin, H, W, dtype=F32, device=x.device) if need_din else None
    dw = torch.zeros_like(weight)
    db = torch.zeros(Cout, dtype=F32, device=x.device)
    ws = _ws(lib().orbit2_conv3x3_bwd_ws_floats, (B, Cin, Cout, H, W), x.device)
    _chk(lib().orbit2_conv3x3_bwd(_p(dout), _p(x), _p(chan_idx), ctot, _p(weight), _p(pre), _p(din), _p(dw), _p(db), B,
                                  Cin, Cout, H, W, mode, r, _p(ws), _stream()), "orbit2_conv3x3_bwd")
    return din, dw, db


def clamp_channel_(img, chan):
    _dev(img, F32, "img")
    B, Cc, H, W = img.shape
    _chk(lib().orbit2_clamp_channel(_p(img), B, Cc, H * W, chan, _stream()), "orbit2_clamp_channel")
    return img


def clamp_channel_bwd_(img_clamped, dimg, chan):
    _dev(img_clamped, F32, "img"); _dev(dimg, F32, "dimg")
    B, Cc, H, W = img_clamped.shape
    _chk(lib().orbit2_clamp_channel_bwd(_p(img_clamped), _p(dimg), B, Cc, H * W, chan, _stream()),
         "orbit2_clamp_channel_bwd")
    return dimg


def loss_fwd(pred, target, lat_w, chan_w, kind):
    _dev(pred, F32, "pred"); _dev(target, F32, "target")
    B, Cc, H, W = pred.shape
    Ht, Wt = target.shape[2], target.shape[3]
    out = torch.empty(Cc + 1, dtype=F32, device=pred.device)
    ws = torch.empty(2 * Cc * B * 64, dtype=F32, device=pred.device)
    _chk(lib().orbit2_loss_fwd(_p(pred), _p(target), Ht, Wt, _p(lat_w), _p(chan_w), _p(out), _p(ws), B, Cc, H, W, kind,
                               _stream()), "orbit2_loss_fwd")
    return out


def loss_bwd(pred, target, lat_w, chan_w, gscale, kind):
    B, Cc, H, W = pred.shape
    Ht, Wt = target.shape[2], target.shape[3]
    dpred = torch.empty_like(pred)
    _chk(lib().orbit2_loss_bwd(_p(pred), _p(target), Ht, Wt, _p(lat_w), _p(chan_w), _p(_dev(gscale, F32, "gscale")),
                               _p(dpred), B, Cc, H, W, kind, _stream()), "orbit2_loss_bwd")
    return dpred


def adamw(p, m, v, g, p16, n, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, found_inf=None):
    _dev(p, F32, "p"); _dev(m, F32, "m"); _dev(v, F32, "v")
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    _chk(lib().orbit2_adamw(_p(p), _p(m), _p(v), _p(g), int(g.dtype == F32), _p(p16), C.c_int64(n), C.c_float(lr),
                            C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(wd), C.c_float(bc1),
                            C.c_float(bc2), C.c_float(grad_scale), _p(found_inf), _stream()), "orbit2_adamw")


def check_finite(g, n, found_inf):
    _chk(lib().orbit2_check_finite(_p(g), int(g.dtype == F32), C.c_int64(n), _p(found_inf), _stream()),
         "orbit2_check_finite")


def selftest(device="cuda") -> int:
    buf = torch.zeros(512, dtype=torch.int32, device=device)
    _chk(lib().orbit2_selftest(_p(buf), _stream()), "orbit2_selftest")
    torch.cuda.synchronize()
    return int(buf[0].item())


def droppath_scales(B, p, seed, device):
    out = torch.empty(B, dtype=F32, device=device)
    _chk(lib().orbit2_droppath_scales(_p(out), B, C.c_float(p), C.c_uint64(seed), _stream()), "orbit2_droppath_scales")
    return out


def transpose_bf16(src, dst):
    _dev(src, BF, "src"); _dev(dst, BF, "dst")
    R, Cc = src.shape
    _chk(lib().orbit2_transpose_bf16(_p(src), _p(dst), R, Cc, _stream()), "orbit2_transpose_bf16")
    return dst


# ---- perceptual loss building blocks (csrc/lpips.hip) ----------------------------------------------------------
def im2col3x3(x, N, H, W, Cc):
    _dev(x, BF, "x")
    col = torch.empty(N * H * W, 9 * Cc, dtype=BF, device=x.device)
    _chk(lib().orbit2_im2col3x3(_p(x), _p(col), N, H, W, Cc, _stream()), "orbit2_im2col3x3")
    return col


def col2im3x3(dcol, N, H, W, Cc, act=None, tapg=None):
    _dev(dcol, BF, "dcol")
    out = torch.empty(N * H * W, Cc, dtype=BF, device=dcol.device)
    _chk(lib().orbit2_col2im3x3(_p(dcol), _p(act), _p(tapg), _p(out), N, H, W, Cc, _stream()), "orbit2_col2im3x3")
    return out


def maxpool2_fwd(x, N, H, W, Cc):
    _dev(x, BF, "x")
    y = torch.empty(N * (H // 2) * (W // 2), Cc, dtype=BF, device=x.device)
    _chk(lib().orbit2_maxpool2_fwd(_p(x), _p(y), N, H, W, Cc, _stream()), "orbit2_maxpool2_fwd")
    return y


def maxpool2_bwd(g, x, N, H, W, Cc, tapg=None):
    _dev(g, BF, "g"); _dev(x, BF, "x")
    dz = torch.empty(N * H * W, Cc, dtype=BF, device=g.device)
    _chk(lib().orbit2_maxpool2_bwd(_p(g), _p(x), _p(tapg), _p(dz), N, H, W, Cc, _stream()), "orbit2_maxpool2_bwd")
    return dz


def lpips_conv1_fwd(img, w1, b1):
    _dev(img, F32, "img"); _dev(w1, F32, "w1"); _dev(b1, F32, "b1")
    N, _, H, W = img.shape
    out = torch.empty(N * H * W, 64, dtype=BF, device=img.device)
    _chk(lib().orbit2_lpips_conv1_fwd(_p(img), _p(w1), _p(b1), _p(out), N, H, W, _stream()), "orbit2_lpips_conv1_fwd")
    return out


def lpips_conv1_bwd(dz, w1, pred, target, l1_coef, gscale=None):
    _dev(dz, BF, "dz"); _dev(pred, F32, "pred"); _dev(target, F32, "target")
    N, _, H, W = pred.shape
    dimg = torch.empty_like(pred)
    gs = None if gscale is None else _p(_dev(gscale, F32, "gscale"))
    _chk(lib().orbit2_lpips_conv1_bwd(_p(dz), _p(w1), _p(pred), _p(target), C.c_float(l1_coef), gs, _p(dimg), N, H, W,
                                      _stream()), "orbit2_lpips_conv1_bwd")
    return dimg


def lpips_tap_fwd(feats, lin, val, B, HW, Cc):
    _dev(feats, BF, "feats"); _dev(lin, F32, "lin"); _dev(val, F32, "val")
    ws = _ws(lib().orbit2_lpips_tap_ws_floats, (B, HW, Cc), feats.device)
    _chk(lib().orbit2_lpips_tap_fwd(_p(feats), _p(lin), _p(val), B, HW, Cc, _p(ws), _stream()), "orbit2_lpips_tap_fwd")


def lpips_tap_bwd(feats, lin, coef, B, HW, Cc, gscale=None):
    _dev(feats, BF, "feats"); _dev(lin, F32, "lin")
    g = torch.empty(B * HW, Cc, dtype=BF, device=feats.device)
    gs = None if gscale is None else _p(_dev(gscale, F32, "gscale"))
    _chk(lib().orbit2_lpips_tap_bwd(_p(feats), _p(lin), _p(g), C.c_float(coef), gs, B, HW, Cc, _stream()),
         "orbit2_lpips_tap_bwd")
    return g


def l1_mean(a, b, out):
    _dev(a, F32, "a"); _dev(b, F32, "b"); _dev(out, F32, "out")
    ws = _ws(lib().orbit2_l1_mean_ws_floats, (C.c_int64(a.numel()),), a.device)
    _chk(lib().orbit2_l1_mean(_p(a), _p(b), _p(out), C.c_int64(a.numel()), _p(ws), _stream()), "orbit2_l1_mean")


def eval_moments(pred, target, lat_w=None, clim=None):
    """[B,C,12] float64 sums over a = pred - clim, b = target - clim (see include/orbit2_hip.h:orbit2_eval_moments)"""
    _dev(pred, F32, "pred"); _dev(target, F32, "target")
    B, Cc, H, W = pred.shape
    if clim is not None:
        _dev(clim, F32, "clim")
        if tuple(clim.shape[-3:]) != (Cc, H, W):
            raise HipBackendError("climatology must be [C,H,W] of the prediction's size")
    out = torch.empty(B, Cc, 12, dtype=torch.float64, device=pred.device)
    _chk(lib().orbit2_eval_moments(_p(pred), _p(target), target.shape[2], target.shape[3], _p(lat_w), _p(clim), _p(out),
                                   B, Cc, H, W, _stream()), "orbit2_eval_moments")
    return out


def seed_salt(value: int, add: bool = False):
    """device-side salt xored into every kernel seed (see include/orbit2_hip.h:orbit2_seed_salt); stream-ordered"""
    _chk(lib().orbit2_seed_salt(C.c_uint64(value & 0xFFFFFFFFFFFFFFFF), int(add), _stream()), "orbit2_seed_salt")
