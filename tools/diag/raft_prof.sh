OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_raft
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/raft_bench.py 2>/dev/null | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $GRAFT_REPO_ROOT/tools/raft_bench.py > $OUT/log 2>&1
f=$(ls $OUT/*kernel_stats.csv $OUT/*/*kernel_stats.csv 2>/dev/null | head -1)
python - <<EOF
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per forward (7 forwards):", tot/1e6/7, "launches per forward:", sum(int(r["Calls"]) for r in rows)/7)
for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"]))[:22]:
    print("%7.3f ms/fwd %6.1f calls/fwd %8.1f us  %s" % (float(r["TotalDurationNs"])/1e6/7, int(r["Calls"])/7, float(r["AverageNs"])/1e3, r["Name"][:90]))
EOF
rm -f $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv
