OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_idle
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --no-host-probe --steps 6 --warmup 3 > $OUT/log 2>&1
f=$(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1)
python $GRAFT_REPO_ROOT/tools/diag/idle_time.py $f
grep "^{" $OUT/log | cut -c1-160
rm -f $f
