# A/B of the row-fused Winograd form on ONE box: step time with VSPW_WINO_ROWS=0 / 1, then the per-shape in-step GEMM
# times (bench.py --kernel-report) of both
mkdir -p gpurun_out/ab
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "winograd or conv2d" 2>&1 | tail -3
for i in 1 2 3; do for r in 0 1; do
  echo "ROWS=$r $(VSPW_WINO_ROWS=$r python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-probe 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], {k:(v["ms_per_step"],v["frac"]) for k,v in d["roofline_hbm"]["families"].items() if "wino_out" in k})')"
done; done
for r in 0 1; do
  VSPW_WINO_ROWS=$r python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-probe --kernel-report gpurun_out/ab/shapes_r${r}.csv > /dev/null 2>&1
  echo "== ROWS=$r"; grep -i "wino" gpurun_out/ab/shapes_r${r}.csv | grep -v wgrad
done
