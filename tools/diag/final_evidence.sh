# Round-end evidence on ONE lease: rocprofv3 passes (tools/gpu_profile.sh) + the bench lines of every configuration
# (tools/diag/final_bench.sh).  Usage (through gpurun): bash tools/diag/final_evidence.sh <tag>
TAG=${1:-r04_final2}
bash tools/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
bash tools/diag/final_bench.sh > gpurun_out/final_bench_$TAG.log 2>&1
tail -40 gpurun_out/final_bench_$TAG.log
