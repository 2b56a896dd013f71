# Re-sweep of the launcher's tuning knobs on ONE lease (interleaved with the default, 2 rounds): ms per step of bench.py.
# usage (through gpurun): bash tools/diag/knob_sweep.sh > gpurun_out/knob_sweep.log
run() { echo "$* : $(env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-probe --no-kernel-timing 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("last_loss"))')"; }
for i in 1 2; do
  run X=default
  run VSPW_WINO_TB=16
  run VSPW_WINO_TB=32
  run VSPW_NBUF1_MAXK=512
  run VSPW_NBUF1_MAXK=2048
  run VSPW_AFFINE_MINC=256
  run VSPW_AFFINE_MINC=1024
  run X=default
  run VSPW_WINO_FUSE_MAXROWS=256
  run VSPW_WINO_FUSE_MAXROWS=1024
  run VSPW_WINO_FUSE_FWD=1
  run VSPW_WINO_ROWS=0
  run VSPW_FWD_APPLY_CONV2=1
  run VSPW_SMALL_WG=260
  run VSPW_SMALL_WG=1040
  run VSPW_WINO_MINC=256
  run X=default
done
