# rocprofv3 kernel trace of the forced-collectives bench (1-rank group): per-kernel time of the SyncBN exchange kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_coll
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VSPW_FORCE_COLLECTIVES=1
for peer in 1 0; do
VSPW_SYNCBN_PEER=$peer rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/peer$peer -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --no-host-probe --steps 4 --warmup 2 > $OUT/peer$peer.log 2>&1
grep "^{" $OUT/peer$peer.log | cut -c1-200
f=$(ls $OUT/peer$peer/*/*kernel_stats.csv $OUT/peer$peer/*kernel_stats.csv 2>/dev/null | head -1)
echo "== $f"; head -1 $f; grep -i "xchg\|nccl\|rccl\|AllReduce\|bn_finalize" $f | cut -c1-220
rm -f $OUT/peer$peer/*/*kernel_trace.csv $OUT/peer$peer/*kernel_trace.csv
done
