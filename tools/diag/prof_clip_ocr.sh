# rocprofv3 kernel stats of the TCB-OCR bench (BASELINE configs[3]): graph replay and serial (eager, side stream off)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_clip_ocr
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --method clip_ocr --no-cpu-baseline --no-kernel-timing --no-host-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B --steps 5 --warmup 2 > $OUT/trace.log 2>&1
VSPW_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -o trace -- $B --mode eager --steps 3 --warmup 2 > $OUT/trace_serial.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $OUT/*/*kernel_trace.csv
for f in $OUT/*.log; do grep "^{" $f | cut -c1-200; done
