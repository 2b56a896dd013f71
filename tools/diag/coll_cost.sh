# step time of the eagerly issued step: no collectives / 1-rank group with the peer exchange / with RCCL all-reduces
B="python bench.py --mode eager --steps 20 --warmup 4 --no-cpu-baseline --no-host-probe --no-kernel-timing"
echo "none: $($B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
for peer in 1 0; do
echo "forced collectives, peer=$peer: $(VSPW_SYNCBN_PEER=$peer VSPW_FORCE_COLLECTIVES=1 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done
echo "forced collectives, no sync bn: $(VSPW_FORCE_COLLECTIVES=1 $B --no-sync-bn 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
