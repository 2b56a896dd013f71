"""GPU idle time inside a graph-replayed step, from a rocprofv3 kernel trace (…_kernel_trace.csv): per step (delimited by
sgd_multi_kernel) the wall time, the union of kernel intervals, the idle remainder and how it is distributed over gaps;
also the union of the MFMA GEMM kernels alone and of everything else alone (overlap = side-stream concurrency)."""
import csv
import sys


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    gaps = []
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            gaps.append(s - ce)
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot, gaps


def main(path):
    rows = list(csv.DictReader(open(path)))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    ks.sort()
    ends = [e for s, e, n in ks if "sgd_multi_kernel" in n]
    print("steps in trace:", len(ends))
    for i in range(max(1, len(ends) - 3), len(ends)):
        lo, hi = ends[i - 1], ends[i]
        step = [(s, e, n) for s, e, n in ks if s >= lo and e <= hi]
        busy, gaps = union([(s, e) for s, e, n in step])
        gemm, _ = union([(s, e) for s, e, n in step if "igemm" in n])
        rest, _ = union([(s, e) for s, e, n in step if "igemm" not in n])
        big = [g for g in gaps if g > 10000]
        print("step %d: wall %.2f ms, %d kernels, busy %.2f ms, idle %.2f ms in %d gaps (median %.1f us, %d gaps > 10 us "
              "= %.2f ms); GEMM union %.2f ms, non-GEMM union %.2f ms, sum of durations %.2f ms"
              % (i, (hi - lo) / 1e6, len(step), busy / 1e6, (hi - lo - busy) / 1e6, len(gaps),
                 sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, len(big), sum(big) / 1e6, gemm / 1e6, rest / 1e6,
                 sum(e - s for s, e, n in step) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
