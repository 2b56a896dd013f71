mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-host-probe --kernel-report gpurun_out/final/gemm_shapes.csv > gpurun_out/final/bench_report.json 2>/dev/null
python bench.py --method clip_ocr --no-cpu-baseline --no-host-probe > gpurun_out/final/bench_clip_ocr.json 2>/dev/null
python tools/eval_bench.py > gpurun_out/final/eval_bench.log 2>&1
python tools/cfg5_bench.py > gpurun_out/final/cfg5_bench.log 2>&1
python tools/netwarp_bench.py > gpurun_out/final/netwarp_bench.log 2>&1
python tools/raft_bench.py > gpurun_out/final/raft_bench.log 2>&1
for f in gpurun_out/final/*.json gpurun_out/final/*.log; do echo "== $f"; tail -2 $f | cut -c1-400; done
head -12 gpurun_out/final/gemm_shapes.csv
