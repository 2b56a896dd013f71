mkdir -p gpurun_out/prof
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/gpu_tests.log
tail -15 gpurun_out/gpu_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -30
tail -2 gpurun_out/prof/bench_under_rocprof.log
