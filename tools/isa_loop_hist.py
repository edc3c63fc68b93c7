"""Instruction histogram of the hottest loop of one kernel in a hipcc -S listing (measurement scaffolding).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o attn.s orbit-2_amd/csrc/attn.hip
    python tools/isa_loop_hist.py attn.s attn_fwd_kernelILi128ELb1ELb0ELi8E [min_mfma]
Loops are found as backward branches (s_cbranch_* / s_branch to an earlier .LBB label); the loop with the most MFMAs is
printed: instruction counts by class, estimated single-wave issue cycles (MI355X_MICROARCH.md cycle constants), registers.
"""
import collections
import re
import sys


def issue_cost(op):
    if op.startswith("v_mfma"):
        return 8
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return 8
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64")):
        return 8
    if op.startswith("v_"):
        return 4
    return 0


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_read:" + op
    if op.startswith("ds_"):
        return "lds_other:" + op
    if op.startswith(("global_load_lds", "buffer_load")) or "lds" in op and op.startswith("global"):
        return "lds_dma"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem:" + op
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu:" + op
    return "other:" + op


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(kern) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] < i:
            loops.append((labels[m.group(2)], i))
    best = None
    for a, b in loops:
        n = sum(1 for l in body[a:b] if re.match(r"^\s+v_mfma", l))
        if best is None or n > best[0] or (n == best[0] and b - a < best[2] - best[1]):
            best = (n, a, b)
    n, a, b = best
    hist = collections.Counter()
    cyc = 0
    scratch = 0
    for l in body[a:b + 1]:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if not m or l.strip().startswith((".", ";")):
            continue
        op = m.group(1)
        hist[classify(op)] += 1
        cyc += issue_cost(op)
        if op.startswith("scratch_"):
            scratch += 1
    print("kernel %s: loop lines %d..%d, %d MFMAs, %d instructions, est. VALU+MFMA issue %d cycles, scratch ops %d"
          % (kern, a, b, n, sum(hist.values()), cyc, scratch))
    groups = collections.Counter()
    for k, v in hist.items():
        groups[k.split(":")[0]] += v
    print("  by class:", dict(groups))
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
        print("   %5d  %s" % (v, k))
    for l in lines[end:end + 80]:
        if re.search(r"(NumVgprs|NumAgprs|ScratchSize|Occupancy|TotalNumVgprs|SpillCount|vgpr_spill|NumSgprs)", l):
            print("  ", l.strip())


if __name__ == "__main__":
    main()
