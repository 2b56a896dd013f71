# tools/diag/ab_env.sh on the TCB-OCR configuration
V=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do for x in "$A" "$B"; do
  echo "$V=$x $(env $V=$x python bench.py --method clip_ocr --steps 20 --warmup 5 --no-cpu-baseline --no-host-probe 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"])')"
done; done
