"""Condenses rocprofv3 CSV output (kernel stats / PMC counter collection) into small text summaries for profiles/."""
import collections
import csv
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[: n.index("(")] if "(" in n else n


def stats(path, out):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats : per-kernel totals (source: %s)\n" % path.split("/")[-1])
        f.write("%-60s %7s %12s %12s %7s\n" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
        for r in rows[:40]:
            f.write("%-60s %7s %12.3f %12.1f %6.1f%%\n" % (short(r["Name"])[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
        f.write("TOTAL %.3f ms over %d kernels\n" % (tot / 1e6, len(rows)))


def pmc(path, out, unit_kb=("FETCH_SIZE", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        k = short(r["Kernel_Name"])
        d = agg.setdefault(k, collections.defaultdict(list))
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out, "a") as f:
        f.write("# rocprofv3 --pmc : per-dispatch averages (source: %s)\n" % path.split("/")[-1])
        for k, d in agg.items():
            if not any(s in k for s in ("gemm", "attn", "ln_", "adamw", "varagg")):
                continue
            f.write(k[:70] + "\n")
            for c, v in d.items():
                avg = sum(v) / len(v)
                extra = ""
                if c == "FETCH_SIZE":
                    extra = "  -> %.1f MB/launch after the gfx950 x2 correction (counter unit KB)" % (avg * 2 * 1024 / 1e6)
                if c == "WRITE_SIZE":
                    extra = "  -> %.1f MB/launch (counter unit KB)" % (avg * 1024 / 1e6)
                f.write("    %-24s avg %16.1f  n=%d%s\n" % (c, avg, len(v), extra))


if __name__ == "__main__":
    mode, src, dst = sys.argv[1:4]
    (stats if mode == "stats" else pmc)(src, dst)
