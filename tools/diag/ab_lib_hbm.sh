# A/B of two library builds on ONE box, with the HBM-family figures: tools/diag/ab_lib_hbm.sh <lib A> <lib B> [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do for l in "$A" "$B"; do
  echo "$(basename $l) $(VSPW_HIP_LIB=$PWD/$l python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-probe 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); f=d["roofline_hbm"]["families"]; print(d["ms_per_step"], {k:(f[k]["avg_launch_ms"], f[k]["frac"]) for k in ("bn_apply","bn_bwd_apply") if k in f})')"
done; done
