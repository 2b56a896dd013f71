"""Summarise a rocprofv3 --pmc pass over tools/conv_probe.py: per kernel name+grid, MFMA-busy fraction and clock."""
import collections, csv, sys
per = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    d = per[r["Dispatch_Id"]]
    d["name"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d["grid"] = r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "")
    d["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    d[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in per.values():
    k = (d["name"], d["grid"])
    a = agg[k]
    a["n"] += 1
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "dur"):
        a[c] += d.get(c, 0.0)
print("%-46s %10s %5s %9s %8s %8s" % ("kernel", "grid", "n", "avg_us", "GHz", "mfma%"))
for (name, grid), a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
    if a["dur"] <= 0 or not name.startswith("igemm"):
        continue
    xcd = a["GRBM_GUI_ACTIVE"] / 8.0
    print("%-46s %10s %5d %9.1f %8.3f %8.1f" % (name[:46], grid, a["n"], a["dur"] / a["n"] / 1e3, xcd / a["dur"],
                                                 100.0 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (xcd * 1024) if xcd else 0))
