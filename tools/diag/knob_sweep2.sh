run() { echo "$* : $(env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-probe --no-kernel-timing 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("last_loss"))')"; }
for i in 1 2 3; do
  run X=default
  run VSPW_NBUF1_MAXK=2048
  run VSPW_NBUF1_MAXK=4096
  run VSPW_NBUF1_MAXK=100000
  run VSPW_NBUF1_MAXK=2048 VSPW_AFFINE_MINC=1024
  run VSPW_AFFINE_MINC=1024
done
