for i in 1 2; do for F in 0 1; do VSPW_DIRECT_FOLD=$F python tools/diag/direct3x3_time.py; done; done
CHUNKS="0" bash tools/diag/parity_chunk.sh > /dev/null 2>&1; grep -E "excess|bench" gpurun_out/parity_chunk.log | sed -e 's/; probs.*//' -e 's/|logit| max [0-9.]*; //' | cut -c1-250
for i in 1 2; do for F in 0 1; do echo "bench DIRECT_FOLD=$F $(VSPW_DIRECT_FOLD=$F python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-probe 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d.get("last_loss"))')"; done; done
