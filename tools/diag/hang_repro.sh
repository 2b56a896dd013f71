#!/bin/bash
# Reproduce the r03 driver-side hang of `bench.py --gpus 2` (shared-GPU gloo test mode) on a fresh lease: run it FIRST
# (cold box), then a few more times; every run under its own timeout, stderr (phase log + watchdog stacks) kept.
mkdir -p gpurun_out/hang
export VSPW_BENCH_SHARED_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-4}
for i in $(seq 1 $N); do
  t0=$(date +%s.%N)
  timeout -s KILL 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-host-probe --no-kernel-timing \
      > gpurun_out/hang/run$i.out 2> gpurun_out/hang/run$i.err
  rc=$?
  t1=$(date +%s.%N)
  echo "run $i rc=$rc $(echo "$t1 - $t0" | bc) s" | tee -a gpurun_out/hang/summary.txt
  tail -n 3 gpurun_out/hang/run$i.out | cut -c1-300
done
