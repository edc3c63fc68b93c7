"""Compact one-character-per-instruction trace of the hottest loop of a kernel (see isa_loop_hist.py for arguments):
M mfma, e transcendental, v other VALU, L lds read, w lds write, D LDS-DMA / global load, S global store, W s_waitcnt,
B s_barrier, s other scalar, | branch / label."""
import re, sys
sys.argv += [""] * 3
from isa_loop_hist import classify   # noqa
def main():
    path, kern = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(kern) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
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
    out = []
    for l in body[a:b + 1]:
        if re.match(r"^\.LBB", l): out.append("|"); continue
        m = re.match(r"^\s+([a-z_0-9]+)(.*)", l)
        if not m or l.strip().startswith((".", ";")): continue
        op = m.group(1)
        if op.startswith("v_mfma"): c = "M"
        elif op.startswith(("v_exp", "v_log", "v_rcp", "v_mul_lo")): c = "e"
        elif op.startswith("v_"): c = "v"
        elif op.startswith("ds_read"): c = "L"
        elif op.startswith("ds_"): c = "w"
        elif op.startswith(("global_load", "buffer_load", "scratch_load")): c = "D"
        elif op.startswith(("global_store", "scratch_store", "buffer_store")): c = "S"
        elif op.startswith("s_waitcnt"):
            c = "W"
            mm = re.search(r"lgkmcnt\((\d+)\)", m.group(2)); vv = re.search(r"vmcnt\((\d+)\)", m.group(2))
            c = "W" + ("l%s" % mm.group(1) if mm else "") + ("v%s" % vv.group(1) if vv else "") + " "
        elif op.startswith("s_barrier"): c = "B"
        elif op.startswith(("s_cbranch", "s_branch")): c = "|"
        else: c = "s"
        out.append(c)
    txt = "".join(out)
    for i in range(0, len(txt), 160): print(txt[i:i + 160])
if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__file__)); main()
