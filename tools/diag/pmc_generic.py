import collections, csv, sys
per = collections.defaultdict(dict)
names = set()
for r in csv.DictReader(open(sys.argv[1])):
    d = per[r["Dispatch_Id"]]
    d["name"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d["grid"] = r.get("Grid_Size", "")
    d["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    d[r["Counter_Name"]] = float(r["Counter_Value"]); names.add(r["Counter_Name"])
names = sorted(names)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in per.values():
    a = agg[(d["name"], d["grid"])]; a["n"] += 1; a["dur"] += d["dur"]
    for c in names: a[c] += d.get(c, 0.0)
print("%-40s %9s %4s %8s " % ("kernel", "grid", "n", "avg_us") + " ".join("%14s" % c[-14:] for c in names))
for (name, grid), a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
    if not name.startswith("igemm"): continue
    print("%-40s %9s %4d %8.1f " % (name[:40], grid, a["n"], a["dur"] / a["n"] / 1e3) + " ".join("%14.4g" % (a[c] / a["n"]) for c in names))
