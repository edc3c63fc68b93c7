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
    assert lib.orbit2_abi_version() == 5


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
