# Shipped library: the raw-weight parity excess and the step time as a function of the accumulation chunk
# (VSPW_ACCUM_CHUNK = 0: one k-sequential chain, the round 1-4 behaviour).  Output: gpurun_out/parity_chunk.log
mkdir -p gpurun_out
OUT=gpurun_out/parity_chunk.log
: > $OUT
K='raw and (cfg2 or cfg3 or cfg4)'
for C in ${CHUNKS:-0 256 128}; do
  echo "=== VSPW_ACCUM_CHUNK=$C" >> $OUT
  VSPW_ACCUM_CHUNK=$C python -m pytest tests/test_fullsize_golden_gpu.py -q -s -k "$K" 2>&1 | grep -E "vs the reference|grad |passed|failed|Error" >> $OUT
done
for i in 1 2; do for C in ${CHUNKS:-0 256 128}; do
  echo "bench VSPW_ACCUM_CHUNK=$C $(VSPW_ACCUM_CHUNK=$C python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-probe 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d.get("last_loss"))')" >> $OUT
done; done
cat $OUT
