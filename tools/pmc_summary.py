#!/usr/bin/env python
"""Summarise the rocprofv3 PMC passes written by tools/gpu_profile.sh into profiles/<tag>_pmc_summary.json:
per-kernel MFMA-busy fraction, effective clock and HBM traffic per launch.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in separate
--pmc passes (FETCH_SIZE costs 3 of the 4 TCC slots), counters are in KiB, and on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is taken as reported (uncalibrated).
"""
import collections
import csv
import json
import sys


def load(path):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d = per[r["Dispatch_Id"]]
        d["name"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d["dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    return per


def group(name):
    if name.startswith("igemm_nt"):  # (incl. igemm_nt_wrows_kernel, the row-fused Winograd GEMM)
        return "igemm_nt_kernel"
    if name.startswith("igemm_tn"):
        return "igemm_tn_kernel"
    return name.split("<")[0]


def main(tag):
    base = "gpurun_out/prof_%s" % tag
    out = {}
    mf = load(base + "/pmc_mfma/pmc_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in mf.values():
        g = agg[group(d["name"])]
        g["launches"] += 1
        for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "dur_ns"):
            g[k] += d.get(k, 0.0)
    for k, g in agg.items():
        if g["dur_ns"] <= 0:
            continue
        xcd_cycles = g["GRBM_GUI_ACTIVE"] / 8.0  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        out[k] = {"launches": int(g["launches"]), "profiled_ms": round(g["dur_ns"] / 1e6, 3),
                  "eff_clock_ghz": round(xcd_cycles / g["dur_ns"], 3),
                  "mfma_busy_frac_of_active_cycles": round(g["SQ_VALU_MFMA_BUSY_CYCLES"] / (xcd_cycles * 1024), 4)
                  if xcd_cycles > 0 else None}
    for cname, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        per = load("%s/%s/pmc_counter_collection.csv" % (base, sub))
        tot = collections.defaultdict(float)
        cnt = collections.Counter()
        for d in per.values():
            tot[group(d["name"])] += d.get(cname, 0.0)
            cnt[group(d["name"])] += 1
        for k in tot:
            if k in out:
                kib = tot[k] / cnt[k]
                out[k][cname + "_KiB_per_launch"] = round(kib, 1)
    for k, v in out.items():
        f, w = v.get("FETCH_SIZE_KiB_per_launch"), v.get("WRITE_SIZE_KiB_per_launch")
        if f is not None and w is not None:
            v["hbm_bytes_per_launch"] = int((2.0 * f + w) * 1024)
    top = dict(sorted(out.items(), key=lambda kv: -kv[1]["profiled_ms"])[:16])
    json.dump({"tag": tag, "method": __doc__.strip().split("\n\n")[1].replace("\n", " "), "kernels": top},
              open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1)
    for k, v in top.items():
        print(k, v)


if __name__ == "__main__":
    main(sys.argv[1])
