# Which layers carry the raw-weight parity excess?  Diagnostic builds that fold the accumulators of the DIRECT 3x3 kernels
# (stem, layer1, the strided 3x3s: the only 3x3s that do not go through Winograd) every N K-tiles, pointwise two-level
# accumulation off / on.  python tools/diag/build_variant.py tapsN -DVSPW_NT_CHUNK_TAPS=N first.  -> gpurun_out/parity_taps.log
mkdir -p gpurun_out
OUT=gpurun_out/parity_taps.log
: > $OUT
K='raw and (cfg2 or cfg3 or cfg4)'
L=$PWD/cvpr2021_vspw_implement_amd/lib
run() { L_=$1; shift; echo "=== $L_" >> $OUT; env "$@" python -m pytest tests/test_fullsize_golden_gpu.py -q -s -k "$K" 2>&1 | grep -E "vs the reference|passed|failed|Error" | sed -e 's/; probs.*//' -e 's/|logit| max [0-9.]*; //' >> $OUT; }
run "direct 3x3: fold every 9 K-tiles (one slab, 288 k); pointwise single chain" VSPW_HIP_LIB=$L/libvspw_hip_taps9.so VSPW_ACCUM_CHUNK=0
run "direct 3x3: fold every 3 K-tiles (96 k); pointwise single chain" VSPW_HIP_LIB=$L/libvspw_hip_taps3.so VSPW_ACCUM_CHUNK=0
run "direct 3x3: fold every K-tile (32 k); pointwise single chain" VSPW_HIP_LIB=$L/libvspw_hip_taps1.so VSPW_ACCUM_CHUNK=0
run "direct 3x3: fold every K-tile (32 k); pointwise chains of 256" VSPW_HIP_LIB=$L/libvspw_hip_taps1.so VSPW_ACCUM_CHUNK=256
run "direct 3x3: fold every K-tile (32 k); pointwise chains of 128" VSPW_HIP_LIB=$L/libvspw_hip_taps1.so VSPW_ACCUM_CHUNK=128
run "direct 3x3: fold every K-tile (32 k); pointwise chains of 64" VSPW_HIP_LIB=$L/libvspw_hip_taps1.so VSPW_ACCUM_CHUNK=64
cat $OUT
