# rocprofv3 passes over the bench workload (run on the GPU box through gpurun); outputs under gpurun_out/prof_<tag>/
TAG=${1:-r01_b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B --steps 3 --warmup 2 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- $B --steps 1 --warmup 1 > $OUT/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $B --steps 1 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $B --steps 1 --warmup 1 > $OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$TAG -type f | head -30
du -sh gpurun_out/prof_$TAG
for f in $OUT/*.log; do echo == $f; grep -v amdgpu.ids $f | tail -2 | cut -c1-300; done
