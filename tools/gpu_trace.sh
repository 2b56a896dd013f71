# the two kernel-trace passes of tools/gpu_profile.sh (no counters) + the per-shape GEMM table; outputs under gpurun_out/prof_<tag>/
TAG=${1:-r03_a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --no-host-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B --steps 5 --warmup 2 > $OUT/trace.log 2>&1
VSPW_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -o trace -- $B --mode eager --steps 3 --warmup 2 > $OUT/trace_serial.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-host-probe --steps 6 --kernel-report $OUT/gemm_shapes.csv > $OUT/report.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python tools/kernel_stats_summary.py $(find $OUT/trace_serial -name "*kernel_stats.csv" | head -1) $OUT/serial_kernel_stats.csv
python tools/kernel_stats_summary.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
du -sh $OUT; for f in $OUT/*.log; do grep "^{" $f | cut -c1-200; done
head -40 $OUT/serial_kernel_stats.csv
