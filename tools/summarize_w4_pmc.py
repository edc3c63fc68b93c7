import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# counter_collection.csv: one row per (dispatch, counter)
d = collections.defaultdict(dict)
for r in rows:
    key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:40], int(r["Grid_Size"]))
    d[key][r["Counter_Name"]] = float(r["Counter_Value"])
    d[key]["t"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) if "End_Timestamp" in r else 0
agg = collections.defaultdict(list)
for (disp, name, grid), c in sorted(d.items()):
    if "gemm256" not in name: continue
    agg[(name, grid)].append(c)
for (name, grid), lst in agg.items():
    lst = lst[len(lst) // 2:]          # second half of each arm: settled clock
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c in lst); act = sum(c["GRBM_GUI_ACTIVE"] for c in lst); t = sum(c["t"] for c in lst)
    # MFMA busy cycles are summed over 4 SIMDs x 256 CUs; GUI_ACTIVE over 8 XCDs
    print("%-42s grid %6d  n=%2d  avg %.3f ms  clock %.2f GHz  matrix pipe busy %.1f %%" % (name, grid, len(lst), t / len(lst) / 1e6, act / 8 / t if t else 0, 100.0 * busy / 1024 / (act / 8)))
