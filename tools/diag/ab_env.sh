# A/B of one environment variable on the default bench: usage ab_env.sh VAR v1 v2 ...   (prints ms/step + HBM families)
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-probe 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
f=d['roofline_hbm']['families']
print('$VAR=$v', d['ms_per_step'], 'NT frac', d['roofline']['frac'], ' '.join('%s %.3fms@%.2f' % (k, x['ms_per_step'], x['frac']) for k, x in f.items()))"
done
