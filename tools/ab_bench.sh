#!/bin/bash
# A/B of two builds of the HIP library on ONE box (box-to-box spread is +-0.5 %): tools/ab_bench.sh <lib A> <lib B> [rounds]
# prints ms/step and the in-bench NT roofline fraction, alternating the libraries.
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for l in "$A" "$B"; do
    echo "$(basename $l) $(VSPW_HIP_LIB=$PWD/$l python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-probe 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"])')"
  done
done
