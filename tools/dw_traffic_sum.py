"""summary of tools/dw_traffic.sh's counter passes: per arm the grouped dW kernel's average duration, effective clock
(GRBM_GUI_ACTIVE / 8 XCDs / duration), matrix-pipe busy share, bytes fetched beyond L2 (FETCH_SIZE x 2 x 1024: gfx950 correction)"""
import csv, glob, os, sys, collections
d = sys.argv[1]
ALG = 2.0 * 131072 * (9216 + 3072 + 3072 + 3072 + 12288 + 3072 + 3072 + 12288) + 2.0 * 3072 * (9216 + 3072 + 12288 + 12288)
for arm in (sys.argv[2:] or ["base", "kwrap8", "kwrap2"]):
    line = "%-9s" % arm
    for kind in ("mfma", "fetch"):
        fs = glob.glob(os.path.join(d, "**", "%s_%s_counter_collection.csv" % (arm, kind)), recursive=True)
        if not fs:
            line += "  (%s pass missing)" % kind
            continue
        disp = collections.defaultdict(dict)
        for r in csv.DictReader(open(fs[0])):
            if "gemm256w_grouped" not in r["Kernel_Name"]:
                continue
            k = int(r["Dispatch_Id"])
            disp[k][r["Counter_Name"]] = float(r["Counter_Value"])
            if "End_Timestamp" in r:
                disp[k]["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        keys = sorted(disp)[-8:]
        n = len(keys)
        if not n:
            line += "  (%s: no dispatch)" % kind
            continue
        ns = sum(disp[k].get("ns", 0.0) for k in keys) / n
        if kind == "mfma":
            gui = sum(disp[k]["GRBM_GUI_ACTIVE"] for k in keys) / n
            busy = sum(disp[k]["SQ_VALU_MFMA_BUSY_CYCLES"] for k in keys) / n
            line += "  %.3f ms  clock %.3f GHz  matrix pipe busy %.1f %%" % (ns / 1e6, gui / 8 / ns if ns else 0, 100 * busy / 1024 / (gui / 8))
        else:
            fb = sum(disp[k]["FETCH_SIZE"] for k in keys) / n * 2 * 1024
            line += "  | fetch pass %.3f ms  FETCH %.1f GB/launch = %.2f x algorithmic (%.1f GB)" % (ns / 1e6, fb / 1e9, fb / ALG, ALG / 1e9)
    print(line)
