#!/usr/bin/env python3
"""Audit of the 4-wave GEMM kernel's register contract (cdna_hip_programming.md §5.7 item 4): its main loop names a[0:255] and
v[128:255] literally, so (1) no compiler-generated v_accvgpr_* may appear outside the asm statements, (2) the kernels must not
use scratch (a spill could land in a named register's lifetime), (3) the kernel descriptor must allocate 256 + 256 registers.
Compiles csrc/gemm.hip to gfx950 assembly and checks every gemm256w kernel.  Exit code 0 = clean."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audit(hipcc="/opt/rocm/bin/hipcc"):
    src = os.path.join(ROOT, "orbit-2_amd", "csrc", "gemm.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "gemm.s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
                            "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stderr[-3000:])
        t = open(out).read()
    found, bad = 0, []
    for m in re.finditer(r"^(_ZN12_GLOBAL__N_1\d+gemm256w\w+):", t, re.M):
        name, i = m.group(1), m.start()
        j = t.index(".Lfunc_end", i)
        body = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", t[i:j], flags=re.S)
        tail = t[j:j + 8000]
        acc = len(re.findall(r"v_accvgpr", body))
        scratch = int(re.search(r"ScratchSize: (\d+)", tail).group(1))
        agpr = int(re.search(r"NumAgprs: (\d+)", tail).group(1))
        vgpr = int(re.search(r"NumVgprs: (\d+)", tail).group(1))
        found += 1
        if acc or scratch or agpr != 256 or vgpr != 256:
            bad.append((name, acc, scratch, agpr, vgpr))
    return found, bad


if __name__ == "__main__":
    found, bad = audit()
    print("%d gemm256w kernels audited" % found)
    for b in bad:
        print("VIOLATION %s: compiler accvgpr %d, scratch %d B, agprs %d, vgprs %d" % b)
    sys.exit(1 if bad or not found else 0)
