"""Condense rocprofv3's trace_kernel_stats.csv into profiles/<tag>_kernel_stats.csv.
The number of training steps in the trace is COUNTED, not assumed: sgd_multi_kernel runs exactly once per step (graph
replays, warm-up and eagerly issued steps alike) - an explicit third argument overrides it."""
import csv, re, sys


def main(src, dst, steps=None):
    rows = list(csv.DictReader(open(src)))
    if steps is None:
        sgd = [int(r["Calls"]) for r in rows if "sgd_multi_kernel" in r["Name"]]
        steps = sgd[0] if sgd else 1
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
    return steps


if __name__ == "__main__":
    print("steps in trace:", main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None))
