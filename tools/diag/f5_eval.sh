# F(5x5,3x3) policy evaluation: step time and full-size parity (raw-weight excess) for a given VSPW_WINO_F5_MINC / _MINCMAX
# usage: bash tools/diag/f5_eval.sh "<minc> <mincmax> [layer-3 tail fraction]" ...
mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg
  export VSPW_WINO_F5_MINC=$1 VSPW_WINO_F5_MINCMAX=$2 VSPW_WINO_F5_L3_TAIL=${3:-0}
  echo "=== F5 where min(c,k) >= $1 and max(c,k) >= $2; layer-3 tail fraction ${3:-0}"
  for i in 1 2; do python bench.py --no-cpu-baseline --no-host-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])"; done
  python -m pytest tests/test_fullsize_golden_gpu.py tests/test_fullsize_gpu.py -q -s 2>&1 | grep -v "^$" | cut -c1-400 > gpurun_out/r06_f5_parity_$1_$2_${3:-0}.log
  tail -1 gpurun_out/r06_f5_parity_$1_$2_${3:-0}.log
  grep -E "^FAILED" gpurun_out/r06_f5_parity_$1_$2_${3:-0}.log | cut -c1-150
  grep -E "excess" gpurun_out/r06_f5_parity_$1_$2_${3:-0}.log | grep -v '"' | sed -e 's/.*(tol/(tol/' -e 's/; probs.*arg-max: / argmax /' -e 's/.*logits/logits/' | cut -c1-150
done
