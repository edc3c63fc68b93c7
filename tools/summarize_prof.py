"""Condenses rocprofv3 CSV output (kernel stats / PMC counter collection) into small text summaries for profiles/."""
import collections
import csv
import os
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


def traffic(fetch_csv, write_csv, out):
    """FETCH_SIZE / WRITE_SIZE passes -> per-launch HBM-side bytes per kernel family (JSON read by bench.py)."""
    import json
    fam = {"gemm": ("gemm256t_kernel", "gemm256t_grouped_kernel", "gemm256w_kernel", "gemm256w_grouped_kernel", "gemm128_kernel",
                    "gemm128_grouped_kernel"),
           "gemm_8phase_single": ("gemm256t_kernel",), "gemm_8phase_grouped_dw": ("gemm256t_grouped_kernel",),
           "gemm_w4_single": ("gemm256w_kernel",), "gemm_w4_grouped_dw": ("gemm256w_grouped_kernel",),
           "attn_fwd": ("attn_fwd_kernel", "attn_fwd_w4_kernel", "attn_fwd_lazy_kernel"),
           "attn_bwd_dq": ("attn_bwd_dq_kernel", "attn_bwd_dq_w4_kernel"), "attn_bwd_dkv": ("attn_bwd_dkv_kernel", "attn_bwd_dkv128_kernel", "attn_bwd_dkv_w4_kernel")}
    acc = {k: {"FETCH_SIZE": [], "WRITE_SIZE": []} for k in fam}
    for path in (fetch_csv, write_csv):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            for f, names in fam.items():
                if any(k.startswith(n) for n in names) and r["Counter_Name"] in acc[f]:
                    acc[f][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    # the whole step: every dispatch of the pass (bench.py --steps 1 --warmup 1 = 2 identical steps, nothing else on the device)
    nsteps = int(os.environ.get("ORBIT2_PMC_STEPS", "2"))
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    for path in (fetch_csv, write_csv):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] in tot:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    res["step"] = {"steps_in_pass": nsteps, "fetch_bytes_per_step": tot["FETCH_SIZE"] * 2 * 1024 / nsteps,
                   "write_bytes_per_step": tot["WRITE_SIZE"] * 1024 / nsteps,
                   "hbm_bytes_per_step": (tot["FETCH_SIZE"] * 2 + tot["WRITE_SIZE"]) * 1024 / nsteps}
    for f, d in acc.items():
        if not d["FETCH_SIZE"]:
            continue
        fb = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 2 * 1024     # gfx950: x2, unit KB
        wb = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) * 1024
        res[f] = {"launches": len(d["FETCH_SIZE"]), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                  "hbm_bytes_per_launch": fb + wb}
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1; FETCH_SIZE "
                    "doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B), counter unit KB; FETCH_SIZE "
                    "is L2-miss traffic (Infinity-Cache hits included)")
    res["_config"] = {"model": "interm_1b", "batch": int(sys.argv[5]) if len(sys.argv) > 5 else 8, "grid": "128x256"}
    # the build the counters belong to: bench.py uses this file only for a library with the same source hash
    sh = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "orbit-2_amd", "lib", "liborbit2_hip.so.srchash")
    res["_srchash"] = open(sh).read().strip() if os.path.exists(sh) else None
    json.dump(res, open(out, "w"), indent=1)


def mall(path, out):
    """TCC_EA0_RDREQ / _LEVEL pass of `bench.py --mall-probe` (or tools/mall_probe.py) -> mean L2-miss latency per kernel of the
    step and the Infinity-Cache hit share, read against FOUR calibration streams run in the same process
    (_hip.mall_calibration: orbit2_probe_read on a 96 MB buffer = Infinity-Cache hits, and on a 4 GB buffer = HBM reads, each
    lightly loaded -- 64 workgroups, one load in flight -- and under a saturating stream -- 2048 workgroups, eight in flight)."""
    import json
    rows = list(csv.DictReader(open(path)))
    per = collections.OrderedDict()                      # dispatch -> counters (dispatch order = launch order)
    for r in rows:
        d = per.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    agg = collections.OrderedDict()
    seen = {"<1>": 0, "<8>": 0}
    for did in sorted(per):
        d = per[did]
        n = d["k"]
        if "probe_read_kernel" in n:
            tag = "<1>" if "<1>" in n else "<8>"
            i = seen[tag]
            seen[tag] += 1
            if tag == "<8>":
                k = None if i == 0 else ("calibration_infinity_cache_saturating" if i <= 6 else "calibration_hbm_saturating")
            else:
                k = "calibration_infinity_cache_light" if i < 6 else "calibration_hbm_light"
            if k is None:
                continue                                 # the sweep that fills the Infinity Cache
        else:
            k = short(n)
            if not any(s_ in k for s_ in ("gemm", "attn_fwd", "attn_bwd")):
                continue
        a_ = agg.setdefault(k, collections.defaultdict(float))
        a_["req"] += d.get("TCC_EA0_RDREQ_sum", 0.0)
        a_["lvl"] += d.get("TCC_EA0_RDREQ_LEVEL_sum", 0.0)
        a_["n"] += 1
    res = collections.OrderedDict()
    for k, a_ in agg.items():
        if a_["req"] > 0:
            res[k] = {"dispatches": int(a_["n"]), "rdreq_per_dispatch": a_["req"] / a_["n"],
                      "mean_l2_miss_latency_cycles": a_["lvl"] / a_["req"]}
    cal = {k[len("calibration_"):]: v["mean_l2_miss_latency_cycles"] for k, v in res.items() if k.startswith("calibration_")}
    if {"infinity_cache_saturating", "hbm_saturating", "infinity_cache_light", "hbm_light"} <= set(cal):
        clip = lambda x: max(0.0, min(1.0, x))
        for k, v in res.items():
            if not k.startswith("calibration"):
                lat = v["mean_l2_miss_latency_cycles"]
                # two interpolations between an Infinity-Cache-hit stream and an HBM stream: endpoints measured lightly loaded
                # (64 workgroups, one load in flight) and under a saturating stream.  The kernel's own fabric load lies between,
                # so the two shares bracket the truth; `_est` is the LOWER one (never flatters the kernel).
                sl = clip((cal["hbm_light"] - lat) / (cal["hbm_light"] - cal["infinity_cache_light"]))
                ss = clip((cal["hbm_saturating"] - lat) / (cal["hbm_saturating"] - cal["infinity_cache_saturating"]))
                v["infinity_cache_hit_share_light_load"] = sl
                v["infinity_cache_hit_share_saturated"] = ss
                v["infinity_cache_hit_share_est"] = min(sl, ss)
                v["within_infinity_cache_bracket"] = bool(cal["infinity_cache_light"] <= lat <= cal["infinity_cache_saturating"])
                v["below_unloaded_hbm_latency"] = bool(lat < cal["hbm_light"])
    res["_calibration"] = cal
    res["_note"] = ("TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum = mean latency of an L2 miss (TCC cycles), kernels measured INSIDE "
                    "the bench step; hit share = linear interpolation between an Infinity-Cache-hit stream and an HBM stream, once "
                    "with lightly loaded and once with saturating calibration streams (orbit2_probe_read); _est = the lower share")
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print(k, v)


def mfma(path, out):
    """SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE pass -> matrix-pipe utilisation per kernel:
    busy cycles summed over the chip's 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs * 1024); the effective clock is
    GRBM_GUI_ACTIVE / 8 / kernel duration (MI355X_MICROARCH.md, DVFS give-back)."""
    rows = list(csv.DictReader(open(path)))
    per = collections.OrderedDict()
    for r in rows:
        d = per.setdefault(r["Dispatch_Id"], {"k": short(r["Kernel_Name"]),
                                              "ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    agg = collections.OrderedDict()
    for d in per.values():
        if "GRBM_GUI_ACTIVE" not in d or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
            continue
        a = agg.setdefault(d["k"], [0.0, 0.0, 0.0, 0])
        a[0] += d["SQ_VALU_MFMA_BUSY_CYCLES"]; a[1] += d["GRBM_GUI_ACTIVE"]; a[2] += d["ns"]; a[3] += 1
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (source: %s)\n" % path.split("/")[-1])
        f.write("%-58s %6s %10s %12s %10s\n" % ("kernel", "calls", "total_ms", "mfma_busy", "clock_GHz"))
        for k, (busy, gui, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
            if busy <= 0:
                continue
            util = busy / (gui / 8.0 * 1024.0)
            f.write("%-58s %6d %10.3f %11.1f%% %10.2f\n" % (k[:58], n, ns / 1e6, 100 * util, gui / 8.0 / ns))


def sq(dirpath, out):
    """every *_counter_collection.csv under dirpath (SQ passes of tools/attn_pmc.sh) -> per kernel (full template name):
    per-dispatch averages of each counter, plus ratios against SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE where they exist."""
    import glob, os
    agg = collections.OrderedDict()
    for path in sorted(glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True)):
        per = {}
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = k[: k.index("(")] if "(" in k else k
            d = per.setdefault((r["Dispatch_Id"], k), {"ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for (_, k), d in per.items():
            a = agg.setdefault(k, collections.defaultdict(list))
            for c, v in d.items():
                a[c].append(v)
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc SQ passes, per-dispatch averages (chip totals); source dir %s\n" % dirpath)
        for k, d in agg.items():
            if not any(s_ in k for s_ in ("gemm", "attn")) or "delta" in k:
                continue
            avg = {c: sum(v) / len(v) for c, v in d.items()}
            f.write("%s   [%.3f ms, n=%d]\n" % (k[:110], avg["ns"] / 1e6, len(d["ns"])))
            wc = avg.get("SQ_WAVE_CYCLES")
            gui = avg.get("GRBM_GUI_ACTIVE")
            for c in sorted(avg):
                if c == "ns":
                    continue
                extra = ""
                if wc and c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_WAVES") and ("CYCLES" in c or "WAIT" in c or "ACTIVE" in c):
                    extra += "  %5.1f %% of SQ_WAVE_CYCLES" % (100 * avg[c] / wc)
                if gui and c == "SQ_VALU_MFMA_BUSY_CYCLES":
                    extra += "  matrix pipe busy %.1f %%, clock %.2f GHz" % (100 * avg[c] / (gui / 8 * 1024), gui / 8 / avg["ns"])
                f.write("    %-34s %18.0f%s\n" % (c, avg[c], extra))


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "sq":
        sq(sys.argv[2], sys.argv[3])
        sys.exit(0)
    if mode == "mfma":
        mfma(sys.argv[2], sys.argv[3])
        sys.exit(0)
    if mode == "mall":
        mall(sys.argv[2], sys.argv[3])
        sys.exit(0)
    if mode == "traffic":
        traffic(*sys.argv[2:5])
    else:
        (stats if mode == "stats" else pmc)(sys.argv[2], sys.argv[3])
