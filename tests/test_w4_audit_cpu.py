"""Register contract of the 4-wave GEMM kernel (csrc/gemm.hip, csrc/gemm_w4_asm.h): its hand-placed main loop owns a[0:255] and
v[128:255] by name, which holds only while the compiler neither touches accumulator registers outside the asm statements nor
spills (tools/audit_w4_asm.py); and the committed instruction stream must be what tools/gen_gemm_w4.py generates."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_generated_loop_is_current(tmp_path):
    out = tmp_path / "gemm_w4_asm.h"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_gemm_w4.py"), "--out", str(out)], check=True,
                   capture_output=True)
    assert out.read_text() == open(os.path.join(ROOT, "orbit-2_amd", "csrc", "gemm_w4_asm.h")).read()


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_compiler_keeps_out_of_the_named_registers():
    import audit_w4_asm
    found, bad = audit_w4_asm.audit()
    assert found >= 8 and not bad, bad
    found, bad = audit_w4_asm.audit_attention()          # the generated attention kernels: one statement = the whole kernel
    assert found == 6 and not bad, bad


def test_generated_attention_streams_are_current(tmp_path):
    for gen, hdr in (("gen_attn_fwd.py", "attn_fwd_asm.h"), ("gen_attn_dq.py", "attn_dq_asm.h"), ("gen_attn_dkv.py", "attn_dkv_asm.h")):
        out = tmp_path / hdr
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen), "--out", str(out)], check=True, capture_output=True)
        assert out.read_text() == open(os.path.join(ROOT, "orbit-2_amd", "csrc", hdr)).read(), hdr
