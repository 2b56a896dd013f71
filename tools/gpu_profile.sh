# rocprofv3 passes over the bench workload (run on the GPU box through gpurun); outputs under gpurun_out/prof_<tag>/
#   trace        : the DEFAULT bench command (hipGraph replay, weight gradients overlapped on the side stream)
#   trace_serial : the same step issued eagerly with the side stream off - every kernel runs alone on the GPU, which is
#                  the condition under which bench.py's live HIP-event roofline leg times igemm_nt_kernel
#   pmc_*        : counter passes on the serial variant (one kernel at a time => counters attribute cleanly)
TAG=${1:-r02_a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --no-host-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B --steps 5 --warmup 2 > $OUT/trace.log 2>&1
export VSPW_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -o trace -- $B --mode eager --steps 3 --warmup 2 > $OUT/trace_serial.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- $B --mode eager --steps 1 --warmup 1 > $OUT/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $B --mode eager --steps 1 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $B --mode eager --steps 1 --warmup 1 > $OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $OUT/*/*kernel_trace.csv.bak
du -sh gpurun_out/prof_$TAG
for f in $OUT/*.log; do echo == $f; grep "^{" $f | cut -c1-240; done
