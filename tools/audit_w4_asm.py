#!/usr/bin/env python3
"""Audit of the register contract of the kernels whose bodies are generated asm statements (cdna_hip_programming.md §5.7 item 4).
4-wave GEMM (csrc/gemm.hip, gemm256w*): the main loop names a[0:255] and v[128:255] literally and the accumulators stay live
across three statements, so OUTSIDE the statements (1) no compiler-generated v_accvgpr_*, (2) no instruction with an accumulator
register operand at all -- on gfx950 loads, stores and LDS instructions take a[...] operands without any v_accvgpr_* -- (the fragment
registers v128..v255 are dead behind the main loop: the compiler may use them between the statements), (3) no scratch,
(4) 256 + 256 registers in the descriptor.
Generated attention kernels (csrc/attn.hip, attn_fwd_w4_kernel / attn_bwd_dq_w4_kernel / attn_bwd_dkv_w4_kernel): the whole body is ONE statement that
ends the kernel: no scratch, 256 + 256 registers, nothing but s_endpgm behind the statement.
Compiles the sources to gfx950 assembly.  Exit code 0 = clean."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asm_of(name, hipcc):
    src = os.path.join(ROOT, "orbit-2_amd", "csrc", name)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
                            "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stderr[-3000:])
        return open(out).read()


def _resources(tail):
    return (int(re.search(r"ScratchSize: (\d+)", tail).group(1)), int(re.search(r"NumAgprs: (\d+)", tail).group(1)),
            int(re.search(r"NumVgprs: (\d+)", tail).group(1)))


ACC_OPERAND = re.compile(r"(?<![\w.])a(\[\d+:\d+\]|\d+)\b")


def audit(hipcc="/opt/rocm/bin/hipcc"):
    t = _asm_of("gemm.hip", hipcc)
    found, bad = 0, []
    for m in re.finditer(r"^(_ZN12_GLOBAL__N_1\d+gemm256w\w+):", t, re.M):
        name, i = m.group(1), m.start()
        j = t.index(".Lfunc_end", i)
        fn = t[i:j]
        body = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", fn, flags=re.S)
        code = "\n".join(l for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";")))
        acc = len(re.findall(r"v_accvgpr", code))
        acc_ops = len(ACC_OPERAND.findall(code))
        scratch, agpr, vgpr = _resources(t[j:j + 8000])
        found += 1
        if acc or acc_ops or scratch or agpr != 256 or vgpr != 256:
            bad.append((name, acc, acc_ops, scratch, agpr, vgpr))
    return found, bad


def audit_attention(hipcc="/opt/rocm/bin/hipcc"):
    t = _asm_of("attn.hip", hipcc)
    found, bad = 0, []
    for m in re.finditer(r"^(_ZN12_GLOBAL__N_1\d+attn_(?:fwd|bwd_dq|bwd_dkv)_w4_kernel\w+):", t, re.M):
        name, i = m.group(1), m.start()
        j = t.index(".Lfunc_end", i)
        fn = t[i:j]
        after = fn[fn.rfind(";;#ASMEND"):].split("\n")[1:]
        trailing = [l.strip() for l in after if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        scratch, agpr, vgpr = _resources(t[j:j + 8000])
        found += 1
        if trailing != ["s_endpgm"] or fn.count(";;#ASMSTART") != 1 or scratch or agpr != 256 or vgpr != 256:
            bad.append((name, trailing[:3], fn.count(";;#ASMSTART"), scratch, agpr, vgpr))
    return found, bad


if __name__ == "__main__":
    found, bad = audit()
    print("%d gemm256w kernels audited" % found)
    for b in bad:
        print("VIOLATION %s: compiler accvgpr %d, accumulator operands outside the statements %d, scratch %d B, agprs %d, vgprs %d" % b)
    fa, ba = audit_attention()
    print("%d generated attention kernels audited" % fa)
    for b in ba:
        print("VIOLATION %s: code behind the statement %s, statements %d, scratch %d B, agprs %d, vgprs %d" % b)
    sys.exit(1 if bad or ba or not found or fa != 6 else 0)
