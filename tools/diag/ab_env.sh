# A/B of one environment switch on ONE box: tools/diag/ab_env.sh VAR A B [rounds]  (bench.py defaults, no CPU baseline)
V=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do for x in "$A" "$B"; do
  echo "$V=$x $(env $V=$x python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-probe 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d["roofline_hbm"]["all"]["ms_per_step"], d.get("last_loss"))')"
done; done
