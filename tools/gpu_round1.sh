mkdir -p gpurun_out/prof
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/gpu_tests.log
tail -6 gpurun_out/gpu_tests.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel-report gpurun_out/kernel_report.csv 2>&1 | tail -1 | tee gpurun_out/bench2.log
head -40 gpurun_out/kernel_report.csv
