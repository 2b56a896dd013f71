#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace into a per-kernel stats CSV
(name, calls, total_us, avg_us, percent) — the `--kernel-trace --stats` summary committed under profiles/."""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:<>]+?)(<.*)?\(", name)
    base = name.split("(")[0]
    if base.startswith("at::native"):
        keys = re.findall(r"(CUDAFunctor_add|FillFunctor|MulFunctor|DivFunctor|direct_copy_kernel_cuda|CatArrayBatchedCopy|"
                          r"uniform_kernel|vectorized_elementwise_kernel|elementwise_kernel_manual_unroll)", name)
        return "torch:" + "/".join(dict.fromkeys(keys)) if keys else "torch:" + base[:60]
    return base


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    agg = {}
    for name, calls, tot, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls
        a[1] += tot
        a[2] += pct
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for k, (calls, tot, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, calls, "%.1f" % tot, "%.3f" % (tot / calls), "%.2f" % pct])
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
