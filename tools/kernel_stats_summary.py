"""Condense rocprofv3's trace_kernel_stats.csv (bench.py --steps 3 --warmup 2 => 5 steps) into profiles/<tag>_kernel_stats.csv."""
import csv, re, sys

def main(src, dst, steps=5):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls(%d steps)" % steps, "total_ms", "avg_us", "percent", "ms_per_step"])
        for r in rows:
            name = re.sub(r"^void ", "", r["Name"])
            name = re.sub(r"\(.*$", "", name)
            if name.startswith("at::native::"):
                name = "torch:" + name[len("at::native::"):][:60]
            tot = float(r["TotalDurationNs"]) / 1e6
            w.writerow([name, r["Calls"], "%.3f" % tot, "%.2f" % (float(r["AverageNs"]) / 1e3), r["Percentage"], "%.3f" % (tot / steps)])

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5)
