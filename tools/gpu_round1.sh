set -x
mkdir -p gpurun_out
python -m pytest tests/test_models_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/models_test.log
tail -30 gpurun_out/models_test.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -5 | tee gpurun_out/bench1.log
