"""CPU checks of the C-ABI library: it builds, loads, and exports every symbol include/orbit2_hip.h declares."""
import ctypes
import os
import re

from tests.conftest import ROOT


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from climate_learn import _hip
    assert os.path.exists(_hip.LIB_PATH)
    lib = ctypes.CDLL(_hip.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "orbit2_hip.h")).read()
    names = sorted(set(re.findall(r"\b(?:int|int64_t)\s+(orbit2_\w+)\s*\(", hdr)))
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert lib.orbit2_abi_version() == 6


def test_gemm_args_struct_matches_header():
    from climate_learn import _hip
    # field order/size contract with the C struct (8-byte pointers, 4-byte ints/floats, 8-byte seed)
    assert ctypes.sizeof(_hip.GemmArgs) == 184
    assert _hip.GemmArgs.colsum_ws.offset == 176          # ABI 5: the last field
    assert _hip.GemmArgs.seed.offset % 8 == 0


def test_no_cpu_fallback():
    import pytest
    import torch
    from climate_learn import _hip
    with pytest.raises(_hip.HipBackendError):
        _hip.layernorm_fwd(torch.zeros(4, 64, dtype=torch.bfloat16), torch.ones(64, dtype=torch.bfloat16),
                           torch.zeros(64, dtype=torch.bfloat16))


def test_gemm_colsum_dispatch_is_host_logic():
    """orbit2_gemm_bf16_colsum_rows mirrors the dispatch without touching the GPU: M / 256 rows for the factor-multiply input gradient
    on whole tiles that fill the chip (the only call that fills colsum_ws), 0 otherwise -- and a call with colsum_ws set that cannot
    fuse is refused (-3) before anything is launched"""
    from climate_learn import _hip
    lib = _hip.lib()

    def args(M, N, K, b_kc=False, mul=True, bias=False, tile=0):
        a = _hip.GemmArgs()
        a.A, a.B, a.C = 0x10000, 0x20000, 0x30000            # never dereferenced on the host
        a.M, a.N, a.K, a.lda, a.ldb, a.ldc = M, N, K, K, (K if b_kc else N), N
        a.a_kc, a.b_kc = 1, int(b_kc)
        a.mul = 0x40000 if mul else None
        a.bias = 0x50000 if bias else None
        a.tile_hint = tile
        return a

    assert lib.orbit2_gemm_bf16_colsum_rows(ctypes.byref(args(131072, 12288, 3072))) == 512      # fc2's input gradient, batch 16
    assert lib.orbit2_gemm_bf16_colsum_rows(ctypes.byref(args(768, 512, 192, tile=260))) == 3
    for bad in (args(768, 512, 192), args(131072, 12288, 3072, b_kc=True), args(131072, 12288, 3072, mul=False),
                args(131072, 12288, 3072, bias=True), args(131072, 12288, 3072, tile=256), args(131000, 12288, 3072)):
        assert lib.orbit2_gemm_bf16_colsum_rows(ctypes.byref(bad)) == 0
        bad.colsum_ws = 0x60000
        assert lib.orbit2_gemm_bf16(ctypes.byref(bad), None) == -3
