mkdir -p gpurun_out
python -m pytest tests/test_sync_gpu.py tests/test_bench_gpu.py tests/test_drivers_gpu.py -m gpu -x -q --timeout 300 -p no:cacheprovider -s --durations=10 > gpurun_out/sync_tests.log 2>&1
tail -25 gpurun_out/sync_tests.log
for peer in 1 0; do
VSPW_SYNCBN_PEER=$peer VSPW_FORCE_COLLECTIVES=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-probe 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['collectives']))"
done
